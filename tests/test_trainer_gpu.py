"""The fused train step (vision_toolbox/trainer.py) against the oracle's restatement of the
harness contract: classifier.py:58-64 (assembly), :91-92 (loss), :111-169 (3-group SGD)."""
import numpy as np
import pytest
import torch

from oracle import filler
from oracle import torch_ref as R
from vision_toolbox import _native as N
from vision_toolbox import backbones
from vision_toolbox.trainer import TrainStep, warmup_cosine_lr

from gpu_util import rel_err

pytestmark = pytest.mark.gpu


def _oracle_steps(name, ncls, x, y, steps, lr, wd, prefix):
    sd = {}
    for k, shape in R.classifier_spec(name, ncls).items():
        dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        sd[k] = filler.fill_tensor(prefix + k, torch.zeros(shape, dtype=dt))
    params = {k: v for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    for v in params.values():
        v.requires_grad_(True)
    mom, losses = {}, []
    for _ in range(steps):
        for v in params.values():
            v.grad = None
        loss, _ = R.classifier_loss(name, sd, x, y, 0.1, training=True)
        loss.backward()
        losses.append(loss.item())
        R.sgd_step(params, {k: v.grad for k, v in params.items()}, mom, lr, 0.9,
                   lambda k: R.weight_decay_group(k, wd, 0.0, 0.0))
    return losses, sd


@pytest.mark.parametrize("name", ["cspdarknet53", "vovnet19_slim_ese"])
@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "hipgraph"])
def test_train_steps_f32_match_oracle(name, graphs):
    # small lr: at 4 images @64px a large step makes the 3-step trajectory chaotic (any two f32
    # implementations diverge), which would test conditioning, not correctness
    ncls, B, S, steps, lr, wd = 16, 8, 96, 3, 2e-4, 1e-3
    x, y = filler.images(B, S), filler.labels(B, ncls)
    ref_losses, ref_sd = _oracle_steps(name, ncls, x, y, steps, lr, wd, "tr.")
    ts = TrainStep(getattr(backbones, name)(), ncls, B, S, torch.float32, lr=lr, momentum=0.9, weight_decay=wd,
                   label_smoothing=0.1, device="cuda", use_graphs=graphs)
    filler.fill_module(ts.model, "tr.")
    ts.weights_changed()
    init = {k: v.detach().clone().cpu() for k, v in ts.model.state_dict().items()}
    before = N.launch_count()
    got = []
    for _ in range(steps):
        ts.step(x.cuda(), y.cuda())
        got.append(ts.loss())
    assert N.launch_count() > before
    # CSPDarknet-53 in train mode is ill-conditioned at this size: f32 rounding reaches ~5e-5 at
    # the last feature map (53 BatchNorm layers on 72 samples per channel), which flips ~1e-4 of
    # the ReLU masks and puts ~2 % noise on every train-mode gradient of ANY f32 implementation
    # (tools/debug_trainer.py; the eval-mode gradient test, free of this, agrees to 1e-4).
    # The 23-unit VoVNet is well conditioned and carries the tight check.
    # (The deep case is a smoke bound, deliberately wide: its trajectory also depends on the ORDER of the
    # f32 atomic adds in the filter gradients, which differs from run to run -- one run in ~12 of the full
    # suite exceeded the earlier 3e-2 / 0.7 / 0.05 bounds.)
    deep = name == "cspdarknet53"
    np.testing.assert_allclose(got, ref_losses, rtol=6e-2 if deep else 1e-2)
    sd = ts.model.state_dict()
    stem = "0.stem.conv.weight" if deep else "0.stem.0.conv.weight"
    for k, tol in ((stem, 0.95 if deep else 0.1), ("3.weight", 0.12 if deep else 0.03), ("3.bias", 0.12 if deep else 0.03)):
        d_got, d_ref = sd[k].cpu() - init[k], ref_sd[k].detach() - init[k]
        assert d_ref.norm() > 0 and rel_err(d_got, d_ref) < tol, k  # the UPDATE, not the weight
    k = [k for k in sd if k.endswith("running_var")][-1]
    assert rel_err(sd[k].cpu(), ref_sd[k]) < (0.12 if deep else 5e-3)


def _oracle_grads(name, ncls, x, y, prefix, training, double=False):
    sd = {}
    for k, shape in R.classifier_spec(name, ncls).items():
        dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        v = filler.fill_tensor(prefix + k, torch.zeros(shape, dtype=dt))
        sd[k] = v.double() if (double and v.is_floating_point()) else v
    params = {k: v for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    for v in params.values():
        v.requires_grad_(True)
    loss, _ = R.classifier_loss(name, sd, x.double() if double else x, y, 0.1, training=training)
    loss.backward()
    return loss.item(), {k: v.grad.float() for k, v in params.items()}


def _device_grads(ts):
    """per-parameter views of the flat f32 gradient buffer the backward list wrote (state_dict naming)."""
    out = {}
    names = {id(p): k for k, p in ts.model.named_parameters()}
    for p, off in zip(ts.store.params, ts.store.offsets):
        g = ts.gflat[off : off + p.numel()]
        g = g.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else g.view(p.shape)
        out[names[id(p)]] = g.detach().cpu().clone()
    return out


def test_cspdarknet53_frozen_bn_step_gradients_match_oracle_tightly():
    """the flagship model's backward wiring under a TIGHT bound.  With running-statistics
    BatchNorm (TrainStep(freeze_bn=True)) every layer is a fixed affine map, so nothing amplifies f32
    rounding and EVERY parameter gradient of the 67-unit model (stem filter gradient, every stride-1 / stride-2
    data gradient, the CSP concat / residual bookkeeping, BatchNorm affine gradients, head) must match the
    float64 oracle: relative L2 error < 2e-3 per parameter, loss to 1e-5.  lr = 0 keeps the weights fixed."""
    name, ncls, B, S = "cspdarknet53", 16, 8, 96
    x, y = filler.images(B, S), filler.labels(B, ncls)
    ref_loss, ref = _oracle_grads(name, ncls, x, y, "trg.", training=False, double=True)
    ts = TrainStep(backbones.cspdarknet53(), ncls, B, S, torch.float32, lr=0.0, momentum=0.0, weight_decay=0.0,
                   label_smoothing=0.1, device="cuda", use_graphs=False, freeze_bn=True)
    filler.fill_module(ts.model, "trg.")
    ts.weights_changed()
    ts.step(x.cuda(), y.cuda())
    assert ts.loss() == pytest.approx(ref_loss, rel=1e-5)
    got = _device_grads(ts)
    assert set(got) == set(ref)
    errs = {k: rel_err(got[k], ref[k]) for k in ref}
    bad = sorted(((e, k) for k, e in errs.items() if not e < 2e-3), reverse=True)
    assert not bad, bad[:8]


def test_cspdarknet53_train_bn_first_step_gradients_track_oracle():
    """Train-mode BatchNorm, step 1 only (no trajectory): per-parameter gradients against the float64 oracle.
    53 BatchNorm layers over 72 samples per channel amplify f32 rounding: the f32 CPU oracle ITSELF sits 2.6 %
    (median over the 203 parameters, max 3.7 %) from the float64 oracle on this input.  The bound is therefore
    relative to that measured conditioning: the HIP path may be no further from float64 than 1.5x (median) / 3x (worst
    parameter) what the f32 CPU oracle is, while a wrong or missing gradient scores >= 1."""
    name, ncls, B, S = "cspdarknet53", 16, 8, 96
    x, y = filler.images(B, S), filler.labels(B, ncls)
    ref_loss, ref = _oracle_grads(name, ncls, x, y, "trh.", training=True, double=True)
    _, cpu32 = _oracle_grads(name, ncls, x, y, "trh.", training=True, double=False)
    ts = TrainStep(backbones.cspdarknet53(), ncls, B, S, torch.float32, lr=0.0, momentum=0.0, weight_decay=0.0,
                   label_smoothing=0.1, device="cuda", use_graphs=False)
    filler.fill_module(ts.model, "trh.")
    ts.weights_changed()
    ts.step(x.cuda(), y.cuda())
    assert ts.loss() == pytest.approx(ref_loss, rel=1e-5)
    got = _device_grads(ts)
    assert set(got) == set(ref)
    errs = sorted(rel_err(got[k], ref[k]) for k in ref)
    base = sorted(rel_err(cpu32[k], ref[k]) for k in ref)
    mid = len(errs) // 2
    assert errs[mid] < 1.5 * base[mid] + 1e-3, (errs[mid], base[mid])
    # (the worst of 203 parameters is one draw from the tail of that noise on either side: a wider factor, still
    #  an order of magnitude below what a wrong gradient would score)
    assert errs[-1] < 3.0 * base[-1] + 1e-3, (errs[-5:], base[-5:])


@pytest.mark.parametrize("pw_min_mb", ["0", "80"], ids=["pointwise_units", "production_threshold"])
def test_bf16_train_step_decreases_loss_and_matches_f32_roughly(pw_min_mb, monkeypatch):
    # VT_PW_MIN_MB is read when a launch list is built: 0 (tests/conftest.py) sends every covered 1x1 unit through the
    # pointwise kernels, 80 is the production default (at this size: the unfused conv + BatchNorm kernels everywhere)
    monkeypatch.setenv("VT_PW_MIN_MB", pw_min_mb)
    ncls, B, S = 16, 8, 64
    x, y = filler.images(B, S), filler.labels(B, ncls)
    losses = {}
    for dt in (torch.float32, torch.bfloat16):
        ts = TrainStep(backbones.cspdarknet53(), ncls, B, S, dt, lr=0.01, device="cuda")
        filler.fill_module(ts.model, "trb.")
        ts.weights_changed()
        ls = []
        for _ in range(6):
            ts.step(x.cuda(), y.cuda())
            ls.append(ts.loss())
        losses[dt] = ls
    assert losses[torch.bfloat16][-1] < losses[torch.bfloat16][0]
    assert losses[torch.bfloat16][0] == pytest.approx(losses[torch.float32][0], rel=3e-2)


def test_lr_schedule_follows_device_scalar():
    ts = TrainStep(backbones.darknet_yolov5n(), 8, 2, 64, torch.float32, lr=0.0, device="cuda")
    w0 = ts.store.pflat.clone()
    ts.step()
    assert torch.equal(w0, ts.store.pflat)  # lr 0: nothing moves
    ts.set_lr(warmup_cosine_lr(3, 100, 0.5))
    ts.step()
    assert not torch.equal(w0, ts.store.pflat)


@pytest.mark.parametrize("sync_bn,model", [("0", "vovnet19_slim_ese"), ("1", "vovnet19_slim_ese"), ("0", "cspdarknet53")],
                         ids=["ddp", "ddp_syncbn", "ddp_cspdarknet53"])
def test_two_ranks_on_one_gpu_match_ddp_semantics(sync_bn, model):
    """bench.py --gpus N path minus RCCL itself: two ranks share this GPU and exchange the gradient
    buckets over gloo (tools/ddp_check.py); updates must equal the oracle's DDP replay and the ranks
    must stay bit-identical.  sync_bn=1: SyncBatchNorm mode against one process on the joint batch."""
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(root / "tools" / "ddp_check.py")],
                       capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DDP_CHECK_SYNCBN=sync_bn, DDP_CHECK_MODEL=model))
    assert r.returncode == 0 and "DDP_CHECK_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    buf = torch.zeros(16, device="cuda")  # the launch-count guard of this module wants a launch here too
    N.check(N.lib().vt_memset(buf.data_ptr(), 0, 64, int(torch.cuda.current_stream().cuda_stream)))


@pytest.mark.parametrize("sync_bn,exchange,collectives",
                         [("0", "allreduce", "torch"), ("1", "allreduce", "torch"), ("0", "sharded", "torch"),
                          ("0", "allreduce", "rccl"), ("1", "allreduce", "rccl")],
                         ids=["rccl", "rccl_syncbn", "rccl_sharded", "inlist", "inlist_syncbn"])
def test_one_rank_rccl_group_runs_the_data_parallel_schedule(sync_bn, exchange, collectives):
    """the N > 1 schedule of TrainStep over a ONE-rank `nccl` (= RCCL) process group on this GPU: bucket all-reduces
    issued from the filter-gradient stream between list segments, SyncBatchNorm collectives, broadcasts -- the calls
    go through torch's RCCL process group exactly as on N GPUs (tools/rccl_world1_check.py); numbers must equal the
    plain schedule's (a one-rank all-reduce is the identity).  `rccl_sharded`: the reduce-scatter -> sharded SGD ->
    all-gather exchange (in-place reduce_scatter_tensor / all_gather_into_tensor on views of the flat buffers);
    `rccl_syncbn` folds the statistics replicas on the device before every all-reduce (vt_stat_fold).  `inlist*` (round 4):
    collectives="rccl" -- the library's own communicator (vt_comm_init over an id torch.distributed carries once), the
    bucket all-reduces and statistics exchanges as ops of the launch lists (VT_OP_ALLREDUCE / VT_OP_STAT_SYNC)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "rccl_world1_check.py"), sync_bn, exchange, collectives],
                       capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    buf = torch.zeros(16, device="cuda")  # the launch-count guard of this module wants a launch here too
    N.check(N.lib().vt_memset(buf.data_ptr(), 0, 64, int(torch.cuda.current_stream().cuda_stream)))


@pytest.mark.parametrize("model,dtype", [("darknet_yolov5n", "f32"), ("vovnet19_slim_ese", "bf16"), ("cspdarknet53", "bf16")])
def test_deterministic_mode_gives_identical_bits(model, dtype):
    """VT_DETERMINISTIC=1 (read once per process, hence the child process): four fresh runs of two SGD steps end with
    bit-identical parameters, gradients, momentum and BatchNorm state (tools/deterministic_check.py) -- the filter
    gradients and bias sums go through a fixed-point shadow like the BatchNorm statistics always do."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "deterministic_check.py"), model, dtype], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, VT_DETERMINISTIC="1"))
    assert r.returncode == 0 and "DETERMINISTIC_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    buf = torch.zeros(16, device="cuda")  # the launch-count guard of this module wants a launch here too
    N.check(N.lib().vt_memset(buf.data_ptr(), 0, 64, int(torch.cuda.current_stream().cuda_stream)))


def test_deterministic_mode_matches_the_oracle_like_the_default_mode():
    """the gradient checks of this module against the float64 / f32 oracle, re-run with VT_DETERMINISTIC=1"""
    import os
    import subprocess
    import sys

    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k",
                        "frozen_bn_step_gradients_match_oracle_tightly or first_step_gradients_track_oracle"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, VT_DETERMINISTIC="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    buf = torch.zeros(16, device="cuda")
    N.check(N.lib().vt_memset(buf.data_ptr(), 0, 64, int(torch.cuda.current_stream().cuda_stream)))


def test_bench_n2_falls_back_together_when_the_library_communicator_cannot_be_made():
    """`bench.py --gpus 2` takes the in-list collectives (the library's own RCCL communicator) on the nccl backend; when the
    communicator cannot be made the ranks AGREE on that and run the torch.distributed form instead.  Two ranks on this one GPU
    over gloo with the rccl branch forced (VT_BENCH_AUTO_RCCL=1): RCCL refuses two ranks on one device (`vt_comm_init: invalid
    usage` on both), both report the fall-back, and the JSON line of the run says `"collectives": "torch"` and carries the
    self-contained N > 1 keys."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, VT_DIST_BACKEND="gloo", VT_FORCE_DEVICE="0", VT_BENCH_AUTO_RCCL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup",
                        "1", "--batch", "16", "--config3-batch", "8"], capture_output=True, text=True, timeout=900, env=env,
                       cwd=str(root))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stderr.count("falling back to torch.distributed") == 2, r.stderr[-3000:]
    line = json.loads(next(l for l in reversed(r.stdout.splitlines()) if l.startswith("{")))
    assert line["n_gpus"] == 2 and line["config"]["collectives"] == "torch" and line["config"]["backend"] == "gloo"
    for key in ("n1_same_per_gpu_batch_ms", "weak_scaling_efficiency", "exchange_exposed_ms"):
        assert key in line
    # BASELINE configs[2] (here: 8 per GPU) measured by the same run, with its own single-GPU denominator
    c3 = line["config3"]
    assert c3["per_gpu_batch"] == 8 and c3["global_batch"] == 16 and c3["ms_per_step"] > 0 and c3["n1_same_per_gpu_batch_ms"] > 0
    assert line["config"]["per_gpu_batch"] == 16 and line["config"]["global_batch"] == 32
    buf = torch.zeros(16, device="cuda")  # the launch-count guard of this module wants a launch here too
    N.check(N.lib().vt_memset(buf.data_ptr(), 0, 64, int(torch.cuda.current_stream().cuda_stream)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("name", ["darknet19", "vovnet19_slim_ese"])
def test_validation_step_matches_the_oracle(name, dtype):
    """TrainStep.validate(): eval-mode forward (running statistics), cross entropy WITHOUT label smoothing, top-1 accuracy --
    `validation_step` of the reference's classifier.py:97-109 -- against the oracle on the same weights and batch, after one
    train step so that the running statistics are not the initial ones.  Nothing is updated by validate()."""
    ncls, B, S = 24, 12, 96
    x, y = filler.images(B, S), filler.labels(B, ncls)
    # (the default TrainStep: launch lists run directly, as bench.py runs them; the captured-graph form is opt-in and has its
    #  own tests above and below)
    ts = TrainStep(getattr(backbones, name)(), ncls, B, S, dtype, lr=1e-3, momentum=0.9, weight_decay=1e-4,
                   label_smoothing=0.1, device="cuda")
    assert ts.use_graphs is False
    filler.fill_module(ts.model, "va.")
    ts.weights_changed()
    ts.step(x.cuda(), y.cuda())
    torch.cuda.synchronize()
    sd = {k: v.detach().clone().cpu() for k, v in ts.model.state_dict().items()}
    before = {k: v.clone() for k, v in sd.items()}
    got = ts.validate(x.cuda(), y.cuda())
    with torch.no_grad():
        logits = R.classifier_logits(name, sd, x, False)
        ref_loss = torch.nn.functional.cross_entropy(logits, y).item()
        ref_hits = int((logits.argmax(-1) == y).sum())
    assert got["count"] == B
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert abs(got["loss"] - ref_loss) <= tol * max(1.0, abs(ref_loss)), (got, ref_loss)
    dev_logits = ts.eval_logits().float().cpu()
    assert rel_err(dev_logits, logits) < (2e-4 if dtype == torch.float32 else 3e-2)
    # top-1: exact against the arg-max of the DEVICE logits (a bf16 near-tie may legitimately differ from the oracle's)
    assert got["correct"] == int((dev_logits.argmax(-1) == y).sum())
    if dtype == torch.float32:
        assert got["correct"] == ref_hits
    assert abs(got["acc"] - got["correct"] / B) < 1e-6
    after = ts.model.state_dict()
    for k, v in before.items():
        assert torch.equal(after[k].cpu(), v), f"validate() changed {k}"
    # a second call repeats (same sums, nothing accumulates across calls)
    again = ts.validate()
    assert again["count"] == B and again["correct"] == got["correct"] and abs(again["loss"] - got["loss"]) < 1e-5


@pytest.mark.parametrize("bucket_mb", ["1", "4"])
def test_captured_data_parallel_segments_equal_the_eager_schedule(bucket_mb):
    """the data-parallel trainer cuts the backward list behind every op that completes a gradient bucket,
    and a cut may land inside a run of held filter gradients (one FORK, up to eight side-stream ops) -- the next segment
    then begins with side-stream ops and no fork.  Captured per segment (use_graphs=True) those launches must still be part
    of the graph: tools/dp_graph_check.py (a one-rank gloo group, VT_DP_WORLD1=1) asserts that such segments exist at this
    bucket size and that every parameter gradient of two replays equals the eager schedule's."""
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "dp_graph_check.py"), bucket_mb], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert r.returncode == 0 and "DP_GRAPH_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    buf = torch.zeros(16, device="cuda")  # the launch-count guard of this module wants a launch here too
    N.check(N.lib().vt_memset(buf.data_ptr(), 0, 64, int(torch.cuda.current_stream().cuda_stream)))


@pytest.mark.parametrize("flag", ["VT_FUSE_BNRED", "VT_BN_BWD_FUSED", "VT_BN_FIN_APPLY"])
def test_fused_batchnorm_forms_leave_the_step_alone(flag, monkeypatch):
    """Round 6 built three launch fusions of the BatchNorm passes: the backward reduction inside the data-gradient launch in
    front of it (VT_FUSE_BNRED) and the whole backward of a unit as one launch with grid barriers (VT_BN_BWD_FUSED) -- OFF by
    default, each measured slower in the step (NOTEBOOK R6.4 / R6.5) -- and the finalize step inside the passes that consume its
    coefficients (VT_BN_FIN_APPLY: every workgroup finalizes for itself; ON by default since R6.10).  Their kernels have parity
    tests of their own; this is the ENGINE side: against the program with all three off (the launches of rounds 1-5), a program
    with one of them on really contains the fused ops, and loss, every parameter gradient and the running statistics of a bf16
    train step are equal (the sums are the same terms in another order; VT_BN_FIN_APPLY is bit-identical)."""
    import ctypes

    ncls, B, S = 16, 8, 96
    x, y = filler.images(B, S), filler.labels(B, ncls)
    want = {"VT_FUSE_BNRED": N.OP_CONV_DGRAD_BNRED, "VT_BN_BWD_FUSED": N.OP_BN_BWD_FUSED, "VT_BN_FIN_APPLY": N.OP_BN_BWD_FIN_APPLY}[flag]

    def run(on):
        for k in ("VT_FUSE_BNRED", "VT_BN_BWD_FUSED", "VT_BN_FIN_APPLY"):
            monkeypatch.setenv(k, "1" if (on and k == flag) else "0")
        ts = TrainStep(backbones.cspdarknet53(), ncls, B, S, torch.bfloat16, lr=0.0, momentum=0.0, weight_decay=0.0,
                       label_smoothing=0.1, device="cuda")
        filler.fill_module(ts.model, "fus.")
        ts.weights_changed()
        ts.step(x.cuda(), y.cuda())
        torch.cuda.synchronize()
        kinds = [ts.prog.bwd_ops[i].kind & 0xFFFF for i in range(ts.prog.n_bwd)]
        rv = {k: v.detach().clone().cpu() for k, v in ts.model.state_dict().items() if k.endswith("running_var")}
        return _device_grads(ts), ts.loss(), kinds.count(want), rv

    g0, l0, n0, rv0 = run(False)
    g1, l1, n1, rv1 = run(True)
    assert n0 == 0 and n1 >= 8, (n0, n1)
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    # (VT_BN_FIN_APPLY is bit-identical; the other two change the ORDER of f32 partial sums, hence the last bf16 bit of a few dz
    #  elements per layer, which 53 train-mode BatchNorm layers amplify on the way down -- 2.1 % at the stem measured, the
    #  decorrelation of DESIGN section 5; a missing or misrouted sum scores ~1)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < (1e-6 if flag == "VT_BN_FIN_APPLY" else 8e-2), k
    for k in rv0:
        assert torch.equal(rv0[k], rv1[k]), k
    n = ctypes.c_uint32(0)
    N.check(N.lib().vt_bn_bwd_fused_timeouts(ctypes.byref(n)))
    assert n.value == 0


def test_grouped_filter_gradients_equal_the_ungrouped_ones(monkeypatch):
    """The engine holds the filter gradients of same-shape 3x3 layers back and releases them as one grouped launch
    (VT_WGRAD_GROUP, default 8; vt_conv_wgrad_group): every parameter gradient of a bf16 train step must equal the ungrouped
    schedule's (VT_WGRAD_GROUP=1) and the in-line form's (VT_WGRAD_INLINE=1) up to the order of the f32 sums, and the grouped
    program must really contain runs of consecutive same-shape filter-gradient ops."""
    ncls, B, S = 16, 8, 96
    x, y = filler.images(B, S), filler.labels(B, ncls)

    def grads(env):
        for k in ("VT_WGRAD_GROUP", "VT_WGRAD_INLINE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ts = TrainStep(backbones.cspdarknet53(), ncls, B, S, torch.bfloat16, lr=0.0, momentum=0.0, weight_decay=0.0,
                       label_smoothing=0.1, device="cuda", use_graphs=False)
        filler.fill_module(ts.model, "grp.")
        ts.weights_changed()
        ts.step(x.cuda(), y.cuda())
        torch.cuda.synchronize()
        ops = ts.prog.bwd_ops
        kinds = [ops[i].kind & 0xFFFF for i in range(ts.prog.n_bwd)]
        runs, cur = [], 0
        for k in kinds:
            cur = cur + 1 if k == N.OP_CONV_WGRAD else 0
            runs.append(cur)
        return _device_grads(ts), max(runs), ts.loss()

    g8, run8, l8 = grads({})
    g1, run1, l1 = grads({"VT_WGRAD_GROUP": "1"})
    gi, runi, li = grads({"VT_WGRAD_INLINE": "1"})
    assert run8 >= 8 and runi >= 8 and run1 <= 2, (run8, runi, run1)
    # (the forward does not depend on the filter-gradient schedule; the reported scalar is a float atomic sum over the rows)
    assert abs(l8 - l1) < 1e-6 * abs(l1) and abs(li - l1) < 1e-6 * abs(l1), (l8, l1, li)
    for k in g8:
        assert rel_err(g8[k], g1[k]) < 1e-5, k
        assert rel_err(gi[k], g1[k]) < 1e-5, k
