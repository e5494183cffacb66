"""The collective entry points of include/vt_amd.h through ctypes alone, one rank (GPU box):

    python tools/comm_abi_check.py

vt_comm_unique_id -> vt_comm_init(rank 0 of 1) -> vt_allreduce_bucket on f32 / bf16 / int64 buffers (a one-rank sum is the
identity: what is checked is that RCCL takes the calls in place on the caller's stream) -> vt_stat_sync on a statistics
buffer with contributions in several replicas (= the fold: replica 0 holds the integer sums, the others are zero) ->
errors: a second vt_comm_init, an unknown dtype, a call after vt_comm_destroy.  No torch.distributed anywhere: this is what a
consumer without PyTorch's process groups would do (the id travels over its own channel).  Prints COMM_ABI_OK."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "vision-toolbox_amd"), str(ROOT)]

import torch

from vision_toolbox import _native as N


def main():
    L = N.lib()
    s = int(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.vt_comm_world() == 0
    buf = torch.arange(1000, device="cuda", dtype=torch.float32)
    assert L.vt_allreduce_bucket(vp(buf), 1000, N.VT_F32, ctypes.c_void_p(s)) == N.VT_ERR_INVALID  # no communicator yet
    ident = ctypes.create_string_buffer(128)
    N.check(L.vt_comm_unique_id(ident))
    assert any(ident.raw)
    N.check(L.vt_comm_init(ident, 0, 1))
    assert L.vt_comm_world() == 1
    assert L.vt_comm_init(ident, 0, 1) == N.VT_ERR_INVALID  # one communicator per process
    for dt, code in ((torch.float32, N.VT_F32), (torch.bfloat16, N.VT_BF16), (torch.int64, N.VT_I64)):
        t = (torch.arange(4099, device="cuda") % 251).to(dt)
        want = t.clone()
        N.check(L.vt_allreduce_bucket(vp(t), t.numel(), code, ctypes.c_void_p(s)))
        torch.cuda.synchronize()
        assert torch.equal(t, want), dt
    assert L.vt_allreduce_bucket(vp(buf), 1000, 7, ctypes.c_void_p(s)) == N.VT_ERR_UNSUPPORTED
    C_ = 40
    stats = N.stats_buffer(C_)
    v0, v1 = torch.randn(C_, device="cuda") * 100, torch.rand(C_, device="cuda") * 1e4
    N.stats_encode(stats, 0, v0 * 0.25, replica=0)
    N.stats_encode(stats, 0, v0 * 0.75, replica=5)
    N.stats_encode(stats, 1, v1, replica=9)
    before = N.stats_decode(stats).clone()
    N.check(L.vt_stat_sync(vp(stats), C_, ctypes.c_void_p(s)))
    torch.cuda.synchronize()
    assert torch.equal(N.stats_decode(stats), before)
    raw = stats.view(torch.int64).reshape(N.VT_STAT_REPLICAS, -1)
    assert int(raw[1:].abs().sum()) == 0 and int(raw[0].abs().sum()) != 0
    N.check(L.vt_comm_destroy())
    assert L.vt_comm_world() == 0
    assert L.vt_allreduce_bucket(vp(buf), 1000, N.VT_F32, ctypes.c_void_p(s)) == N.VT_ERR_INVALID
    print("COMM_ABI_OK")


if __name__ == "__main__":
    main()
