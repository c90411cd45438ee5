"""Trainer (Keras-flavour flow) and Estimator (TF-flavour flow) on CPU."""
import torch

import distributeddeeplearning_amd.torch as hvd
from distributeddeeplearning_amd import callbacks as cb
from distributeddeeplearning_amd.models import resnet
from distributeddeeplearning_amd.optim import FusedSGD
from distributeddeeplearning_amd.training import (BroadcastGlobalVariablesHook, Estimator,
                                                  EstimatorSpec, ModeKeys, Trainer)
from distributeddeeplearning_amd.utils import checkpoint, get_logger


def _batches(n, bs=4, classes=5, one_hot=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.randn(bs, 3, 16, 16, generator=g)
        y = torch.randint(0, classes, (bs,), generator=g)
        out.append((x, torch.nn.functional.one_hot(y, classes).float() if one_hot else y))
    return out


def test_trainer_fit_evaluate_with_callbacks(tmp_path):
    hvd.init()
    torch.manual_seed(0)
    model = resnet(18, num_classes=5)
    opt = hvd.DistributedOptimizer(FusedSGD(model.named_parameters(), lr=0.004, momentum=0.9),
                                   named_parameters=model.named_parameters())
    data = _batches(3, one_hot=True)  # Keras iterators deliver one-hot labels
    hooks = [cb.BroadcastGlobalVariablesCallback(0), cb.MetricAverageCallback(),
             cb.LearningRateWarmupCallback(warmup_epochs=1, size=4),
             cb.LoggerCallback(get_logger("t"), data_length=12),
             cb.CheckpointCallback(str(tmp_path))]
    tr = Trainer(model, opt, callbacks=hooks)
    hist = tr.fit(data, epochs=3, steps_per_epoch=3)
    assert [h["epoch"] for h in hist] == [0, 1, 2]
    assert hist[-1]["loss"] < hist[0]["loss"]
    assert set(hist[0]) >= {"loss", "acc", "top_k_categorical_accuracy", "lr"}
    assert checkpoint.latest_epoch(str(tmp_path)) == 3
    scores = tr.evaluate(data)
    assert len(scores) == 3 and 0.0 <= scores[1] <= scores[2] <= 1.0
    assert tr.predict(data, steps=1).shape == (4, 5)
    # resume: initial_epoch skips finished epochs
    tr2 = Trainer(model, opt)
    assert tr2.fit(data, epochs=3, initial_epoch=3) == []


def _model_fn(features, labels, mode, params):
    st = params["state"]
    if "model" not in st:
        torch.manual_seed(1)
        st["model"] = resnet(18, num_classes=5)
        st["optimizer"] = FusedSGD(st["model"].named_parameters(), lr=0.05, momentum=0.9)
    model, opt = st["model"], st["optimizer"]
    if mode == ModeKeys.PREDICT:
        model.eval()
        return EstimatorSpec(mode, predictions={"classes": model(features).argmax(1)})
    model.train(mode == ModeKeys.TRAIN)
    logits = model(features)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    if mode == ModeKeys.EVAL:
        return EstimatorSpec(mode, loss=loss, eval_metric_ops={"accuracy": (logits.argmax(1) == labels).float().mean()})

    def train_op():
        opt.zero_grad()
        loss.backward()
        opt.step()

    return EstimatorSpec(mode, loss=loss, train_op=train_op)


def test_estimator_train_eval_predict_and_model_dir(tmp_path):
    hvd.init()
    data = _batches(4)
    est = Estimator(_model_fn, model_dir=str(tmp_path), params={})
    hook = BroadcastGlobalVariablesHook(0)
    before = est.evaluate(lambda: iter(data)) if False else None  # model is created lazily by train
    est.train(lambda: iter(data * 3), steps=10, hooks=[hook])
    assert est.global_step == 10 and hook.done
    ev = est.evaluate(lambda: iter(data))
    assert set(ev) == {"loss", "accuracy", "global_step"} and ev["global_step"] == 10
    preds = list(est.predict(lambda: iter(data)))
    assert len(preds) == 4 and preds[0]["classes"].shape == (4,)
    assert checkpoint.latest_epoch(str(tmp_path)) == 10  # rank-0 model_dir checkpoint
    # a new estimator on the same model_dir restores and continues the step count
    est2 = Estimator(_model_fn, model_dir=str(tmp_path), params={})
    est2.train(lambda: iter(data), steps=2)
    assert est2.global_step == 12
    assert before is None


def test_graphed_train_step_runs_eagerly_without_a_gpu():
    """training/graph.py on CPU tensors: never captures, behaves exactly like the plain loop
    (zero_grad, forward, loss, backward, step) and returns a detached loss."""
    import torch
    from distributeddeeplearning_amd.training.graph import GraphedTrainStep

    def make():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        return m, torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)

    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(5, 8, generator=g) for _ in range(6)]
    ts = [torch.randint(0, 4, (5,), generator=g) for _ in range(6)]
    loss_fn = torch.nn.functional.cross_entropy
    m1, o1 = make()
    ref = []
    for x, t in zip(xs, ts):
        o1.zero_grad()
        loss = loss_fn(m1(x), t)
        loss.backward()
        o1.step()
        ref.append(float(loss))
    m2, o2 = make()
    step = GraphedTrainStep(m2, o2, loss_fn, warmup=2)
    got = []
    for x, t in zip(xs, ts):
        out = step(x, t)
        assert not out.requires_grad
        got.append(float(out))
    assert not step.graphed and step.failed is None
    assert got == ref
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_build_graph_step_policy():
    """training.build_graph_step: never on a CPU device; on a GPU ``auto`` means one rank and a
    per-GPU batch of at most 32, ``on`` / ``off`` force it (the device is only inspected by type)."""
    import torch
    from distributeddeeplearning_amd.training import build_graph_step
    from distributeddeeplearning_amd.training.graph import GraphedTrainStep
    from distributeddeeplearning_amd.utils import TrainConfig

    cpu, gpu = torch.device("cpu"), torch.device("cuda", 0)
    mk = lambda mode, bs: TrainConfig(hip_graph=mode, batch_size=bs)  # noqa: E731
    assert build_graph_step(mk("on", 16), None, None, None, cpu, 1) is None
    assert build_graph_step(mk("off", 16), None, None, None, gpu, 1) is None
    assert build_graph_step(mk("auto", 64), None, None, None, gpu, 1) is None
    assert build_graph_step(mk("auto", 16), None, None, None, gpu, 2) is None
    assert isinstance(build_graph_step(mk("auto", 32), None, None, None, gpu, 1), GraphedTrainStep)
    assert isinstance(build_graph_step(mk("on", 256), None, None, None, gpu, 8), GraphedTrainStep)
