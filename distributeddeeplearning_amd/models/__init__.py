from .resnet import (ResNet, resnet, resnet18, resnet34, resnet50, resnet101, resnet152,
                     resnet200, DEPTHS, count_parameters)


def build_model(name: str, **kw):
    """``build_model("resnet50")`` — the reference selects its model by name
    (``models.__dict__["resnet50"]``, imagenet_pytorch_horovod.py:323)."""
    table = {"resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50,
             "resnet101": resnet101, "resnet152": resnet152, "resnet200": resnet200}
    if name not in table:
        raise KeyError("unknown model %r (have %s)" % (name, sorted(table)))
    return table[name](**kw)
