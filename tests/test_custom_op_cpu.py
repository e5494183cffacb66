"""The operator the module API dispatches to, `torch.ops.vision_toolbox_amd.backbone` (torch.library
registration with a fake-tensor shape function, so that tracing works; reference tests/test_backbones.py:76-86).
CPU-side checks: registration, schema, and the fake (meta) implementation on fake CUDA tensors -- no GPU needed."""
import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode

from vision_toolbox import _native as N
from vision_toolbox import backbones
from vision_toolbox import program as P


def test_operator_is_registered_with_a_cuda_kernel_and_autograd():
    op = torch.ops.vision_toolbox_amd.backbone.default
    schema = str(op._schema)
    assert "Tensor x" in schema and "Tensor[] params" in schema and "-> Tensor[]" in schema
    # a CPU tensor never reaches the operator (dispatch rule): calling it directly on one fails loudly
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.vision_toolbox_amd.backbone(torch.zeros(1, 3, 32, 32), [], 0, True, N.VT_F32, False)


@pytest.mark.parametrize("name,all_maps,dtype", [("darknet19", True, N.VT_F32), ("cspdarknet53", False, N.VT_BF16),
                                                 ("vovnet19_slim_ese", True, N.VT_BF16)])
def test_fake_implementation_gives_shapes_strides_and_dtypes(name, all_maps, dtype):
    m = getattr(backbones, name)()
    runner = m._vt_runner() if hasattr(m, "_vt_runner") else None
    if runner is None:
        pytest.skip("backbone has no runner accessor")
    runner.store.ensure(torch.device("cpu"))  # layout only; the fake implementation touches no data
    with FakeTensorMode():
        x = torch.empty(2, 3, 64, 64, device="cuda")
        outs = torch.ops.vision_toolbox_amd.backbone(x, [], runner.handle, all_maps, dtype, False)
    maps, token = outs[:-1], outs[-1]
    chans = m.out_channels_list if all_maps else m.out_channels_list[-1:]
    assert len(maps) == len(chans)
    assert token.dtype == torch.int64 and token.device.type == "cpu"
    want_dtype = torch.float32 if dtype == N.VT_F32 else torch.bfloat16
    for t, c in zip(maps, chans):
        assert t.shape[0] == 2 and t.shape[1] == c and t.dtype == want_dtype and t.device.type == "cuda"
        assert t.is_contiguous(memory_format=torch.channels_last)
    assert maps[-1].shape[-1] == 64 // m.stride


def test_unknown_handle_raises():
    with FakeTensorMode():
        x = torch.empty(1, 3, 32, 32, device="cuda")
        with pytest.raises(RuntimeError, match="no longer exists"):
            torch.ops.vision_toolbox_amd.backbone(x, [], 1 << 40, True, N.VT_F32, False)
