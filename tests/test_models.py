"""L2 models: ResNet family topology, torchvision-compatible state_dict."""
import pytest
import torch

from distributeddeeplearning_amd.models import build_model, count_parameters, resnet, resnet50


def test_resnet50_parameter_count_and_tensors():
    m = resnet50()
    assert count_parameters(m) == 25557032
    assert len(list(m.parameters())) == 161
    assert len(m.state_dict()) == 320  # 161 params + 53 x 3 BN buffers


def test_resnet152_parameter_count():
    assert count_parameters(resnet(152)) == 60192808


def test_state_dict_keys_match_torchvision_naming():
    sd = resnet50().state_dict()
    for k in ("conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var",
              "bn1.num_batches_tracked", "layer1.0.conv1.weight", "layer1.0.bn3.weight",
              "layer1.0.downsample.0.weight", "layer1.0.downsample.1.running_var",
              "layer4.2.conv3.weight", "fc.weight", "fc.bias"):
        assert k in sd, k
    assert tuple(sd["conv1.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["layer2.0.conv2.weight"].shape) == (128, 128, 3, 3)
    assert tuple(sd["fc.weight"].shape) == (1000, 2048)
    assert sd["bn1.num_batches_tracked"].dtype == torch.int64
    assert "layer1.1.downsample.0.weight" not in sd  # projection only in the first block


@pytest.mark.parametrize("depth,feat", [(18, 512), (34, 512), (50, 2048)])
def test_forward_backward_cpu(depth, feat):
    m = resnet(depth, num_classes=10)
    assert m.feature_dim == feat
    x = torch.randn(2, 3, 32, 32)
    y = m(x)
    assert y.shape == (2, 10)
    y.sum().backward()
    assert all(p.grad is not None for p in m.parameters())
    assert int(m.bn1.num_batches_tracked) == 1


def test_stride_on_3x3_and_zero_init():
    m = resnet50(zero_init_residual=True)
    blk = m.layer2[0]
    assert blk.conv1.stride == 1 and blk.conv2.stride == 2  # v1.5 (resnet_model.py:188-204)
    assert float(blk.bn3.weight.abs().sum()) == 0.0
    assert float(resnet50().layer2[0].bn3.weight.sum()) == 512.0


def test_build_model_by_name():
    assert build_model("resnet18", num_classes=5).fc.weight.shape[0] == 5
    with pytest.raises(KeyError):
        build_model("vgg16")
    with pytest.raises(ValueError):
        resnet(51)


def test_eval_mode_uses_running_stats():
    m = resnet(18, num_classes=4).eval()
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        a, b = m(x), m(x)
    assert torch.equal(a, b) and int(m.bn1.num_batches_tracked) == 0
