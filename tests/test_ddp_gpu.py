"""The data-parallel PRODUCT step with world_size = 2 on one MI355X: two processes share cuda:0 and exchange gradients through a gloo group on
device tensors (RCCL refuses two ranks on one device; the collectives, their order, the overlapped per-block schedule from the real backward
hooks and the flat-buffer bookkeeping are the same code - only the transport differs).  Multi-GPU RCCL runs are the driver's (bench.py --gpus N)."""

from __future__ import annotations

import math
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from conftest import ROOT  # noqa: E402


def _kwargs() -> dict:
    views = ["sax", "lax_2c"]
    return dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
                enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
                enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2, dec_n_heads=4)


def _worker(rank: int, world: int, port: int, tmp: str) -> None:
    for p in (str(ROOT), str(ROOT / "oracle")):
        sys.path.insert(0, p)
    import cinema_oracle as O  # noqa: N812  (mask recipe only)
    from cinema_amd import CineMA
    from cinema_amd.ddp import GradientSynchronizer, ddp_setup
    from cinema_amd.optim import FlatModel, TrainStep

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.cuda.set_device(0)
    ddp_setup(rank, world, port=port, backend="gloo")
    kw = _kwargs()
    cfg = O.MAEConfig(**kw)
    gen = torch.Generator().manual_seed(3)
    images = {v: torch.rand(4, 1, *s, generator=gen) for v, s in kw["image_size_dict"].items()}
    masks = {v: O.random_patch_mask(4, math.prod(cfg.grid_size(v)), 0.75, gen) for v in images}
    half = slice(2 * rank, 2 * rank + 2)

    # (a) eager step, injected masks: mean over ranks of the half-batch gradients == full-batch gradient, through the REAL backward hooks
    torch.manual_seed(100 + rank)  # different initial weights per rank: attach() must broadcast rank 0's
    model = CineMA(**kw).to("cuda")
    flat = FlatModel(model, 0.05)
    sync = GradientSynchronizer(world)
    sync.min_early = 1 << 12  # the test model's blocks are small: let the overlapped per-block collectives fire
    sync.attach(flat)
    loss, _, _, _ = model({v: images[v][half].cuda() for v in images}, 0.75, enc_mask_dict={v: masks[v][half].cuda() for v in images})
    sync.arm(True)
    loss.backward()
    sync.all_reduce()
    torch.cuda.synchronize()
    torch.save({"grad": flat.flat_grad.cpu(), "param": flat.flat_param.cpu(), "n_early": sync.n_early_last, "loss": float(loss)}, f"{tmp}/a{rank}.pt")
    if rank == 0:  # single-process reference on the full batch, same weights (rank 0's), no synchroniser
        from cinema_amd import tape as T

        hook, T.PARAMS_DONE_HOOK = T.PARAMS_DONE_HOOK, None
        ref = CineMA(**kw)
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        ref.to("cuda")
        rflat = FlatModel(ref, 0.05)
        rloss, _, _, _ = ref({v: images[v].cuda() for v in images}, 0.75, enc_mask_dict={v: masks[v].cuda() for v in images})
        rloss.backward()
        torch.cuda.synchronize()
        torch.save({"grad": rflat.flat_grad.cpu(), "loss": float(rloss)}, f"{tmp}/full.pt")
        T.PARAMS_DONE_HOOK = hook
    torch.distributed.barrier()

    # (b) the way bench.py runs: TrainStep(replay=True) + synchroniser, rank-local data and random masks; the ranks must stay bit-identical
    torch.manual_seed(7 + rank)
    model2 = CineMA(**kw).to("cuda")
    sync2 = GradientSynchronizer(world)
    sync2.min_early = 1 << 12
    step = TrainStep(model2, lr=1e-3, synchronizer=sync2, replay=True)
    torch.manual_seed(1000 + rank)  # per-rank mask / data streams (cinema/mae/pretrain.py:309-310)
    batches = [{v: torch.rand(2, 1, *s, device="cuda") for v, s in kw["image_size_dict"].items()} for _ in range(2)]
    early, losses = [], []
    for i in range(5):
        loss, gn, _ = step(batches[i % 2], 0.75)
        losses.append((float(loss), float(gn)))
        early.append(sync2.n_early_last)
    rec = next(iter(step._recorded.values()))  # noqa: SLF001
    torch.cuda.synchronize()
    torch.save({"param": step.flat.flat_param.cpu(), "losses": losses, "early": early, "host_entries": sum(1 for fn, _ in rec.calls if fn is None)},
               f"{tmp}/b{rank}.pt")
    torch.distributed.destroy_process_group()


def test_two_rank_product_step_on_one_gpu(tmp_path: Path) -> None:
    from cinema_amd.ddp import get_free_port

    mp.spawn(_worker, args=(2, get_free_port(), str(tmp_path)), nprocs=2, join=True)
    a0, a1, full = (torch.load(tmp_path / f) for f in ("a0.pt", "a1.pt", "full.pt"))
    assert torch.equal(a0["param"], a1["param"])                    # rank 0's weights everywhere
    assert torch.equal(a0["grad"], a1["grad"])                      # every rank holds the same (mean) gradient
    assert a0["n_early"] >= 2 and a0["n_early"] == a1["n_early"]    # per-block collectives were issued from the backward hooks, matched across ranks
    rel = float((a0["grad"] - full["grad"]).norm() / full["grad"].norm())
    assert rel <= 2e-2, rel                                         # 2-rank split batch == 1-rank full batch (bf16 compute, different tile shapes)
    assert abs(0.5 * (a0["loss"] + a1["loss"]) - full["loss"]) <= 2e-3 * abs(full["loss"])
    b0, b1 = torch.load(tmp_path / "b0.pt"), torch.load(tmp_path / "b1.pt")
    assert torch.equal(b0["param"], b1["param"])                    # replayed steps with overlapped exchange: replicas stay bit-identical
    assert b0["host_entries"] >= 2 and b0["early"] == b1["early"] and min(b0["early"]) >= 2  # the hooks are host entries of the recording and fire on every replay
    assert [g for _, g in b0["losses"]] == [g for _, g in b1["losses"]]  # the pre-clip norm is computed from the reduced gradient: identical
    assert b0["losses"] != b1["losses"]                             # different data per rank: different local losses
    assert all(math.isfinite(l) and math.isfinite(g) for l, g in b0["losses"])


@pytest.mark.parametrize(("exchange", "algorithm", "launcher"), [("fp32", "all_reduce", "torchrun"), ("bf16", "all_reduce", "torchrun"), ("fp32", "rs_ag", "torchrun"),
                                                                 ("fp32", "all_reduce", "plain")])
def test_bench_two_ranks_on_one_gpu_reports_the_ddp_block(exchange: str, algorithm: str, launcher: str) -> None:
    """``bench.py --gpus 2`` exactly as the driver launches it (``python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2``), with both ranks on the
    one leased GPU and gloo as the transport (dev overrides CINEMA_BENCH_SHARE_GPU / CINEMA_BENCH_BACKEND: RCCL refuses two ranks on one device; the rank
    environment, barrier, max-over-ranks timing, rank-0 JSON line, replayed step + GradientSynchronizer are the product's).  Checks the JSON contract of the
    N > 1 line and its ``ddp`` object (all-reduce in fp32 and bf16, and the reduce-scatter + all-gather exchange): both ranks seen, payload = every element of the flat gradient buffer once per step (4 bytes fp32, 2 bytes bf16),
    overlapped per-block collectives issued, the three schedule timings present."""
    import json
    import subprocess

    from cinema_amd import CineMA
    from cinema_amd.ddp import get_free_port

    sys.path.insert(0, str(ROOT))
    import bench

    env = dict(os.environ, CINEMA_BENCH_SHARE_GPU="1", CINEMA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # launcher "plain": `python bench.py --gpus 2` with no launcher and no rank environment (the form the driver's BENCH command uses): bench.py re-executes itself
    # under torch.distributed.run (bench.self_launch)
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")} if launcher == "plain" else env
    head = [sys.executable] if launcher == "plain" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                         "--master-port", str(get_free_port())]
    cmd = [*head, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm", "1", "--batch", "2", "--cpu-budget", "0", "--grad-exchange", exchange, "--exchange", algorithm]
    import signal
    import tempfile

    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:  # files, not pipes: a killed launcher's children cannot keep a pipe open
        proc = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, text=True, cwd=str(ROOT), start_new_session=True)
        try:
            rc = proc.wait(timeout=240)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)  # the whole process group: launcher + both ranks
            proc.wait()
            fe.seek(0)
            pytest.fail("bench.py --gpus 2 did not finish within 240 s: " + fe.read()[-3000:])
        fo.seek(0)
        fe.seek(0)
        stdout, stderr = fo.read(), fe.read()
    assert rc == 0, stderr[-3000:]
    line = [ln for ln in stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) <= 0.02 * d["value"]  # whole-job samples/s over both ranks
    ddp = d["ddp"]
    model = CineMA(**bench.base_kwargs("base"))
    flat_elems = sum(((p.numel() + 7) // 8) * 8 for p in model.parameters() if p.requires_grad)  # FlatModel pads every parameter to 8 elements
    assert ddp["n_ranks_seen"] == 2 and ddp["backend"] == "gloo" and ddp["exchange_dtype"] == exchange
    assert ddp["payload_bytes_per_step"] == flat_elems * (4 if exchange == "fp32" else 2), (ddp["payload_bytes_per_step"], flat_elems)
    assert ddp["early_collectives_per_step"] >= 20  # 12 encoder + 8 decoder blocks (+ the shared k|v range) go out from the backward hooks
    # the explicit reduce-scatter + all-gather exchange (--exchange rs_ag): two collectives per range, same payload; the RCCL / NCCL knobs in force are echoed
    assert ddp["exchange_algorithm"] == algorithm and isinstance(ddp["rccl_env"], dict)
    assert ddp["collectives_per_step"] >= (2 if algorithm == "rs_ag" else 1) * ddp["early_collectives_per_step"]
    assert all(ddp[k] > 0 for k in ("ms_per_step_overlapped", "ms_per_step_exchange_after_backward", "ms_per_step_no_exchange"))


@pytest.mark.parametrize(("exchange", "algorithm"), [("fp32", "all_reduce"), ("bf16", "rs_ag")])
def test_bench_one_rank_over_rccl_issues_the_overlapped_collectives(exchange: str, algorithm: str) -> None:
    """``bench.py --force-sync``: a ONE-rank process group on the real RCCL backend ("nccl") - the only way RCCL itself runs on a one-GPU box.  The overlapped per-block
    collectives (all_reduce, and reduce_scatter_tensor + all_gather_into_tensor) are issued from the weight-gradient stream's hooks inside the replayed step exactly as
    they are for N > 1; with one rank the mean is the identity, so the loss must equal the plain single-process run's."""
    import json
    import subprocess

    def run(extra: list) -> dict:
        cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--prewarm", "1", "--batch", "2", "--cpu-budget", "0", "--profile-steps", "0", "--no-secondary", *extra]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, check=False, cwd=str(ROOT), env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])

    plain = run([])
    rccl = run(["--force-sync", "--grad-exchange", exchange, "--exchange", algorithm])
    assert rccl["n_gpus"] == 1 and rccl["config"]["final_loss"] > 0
    ddp = rccl["ddp"]
    assert ddp["backend"] == "nccl" and ddp["exchange_algorithm"] == algorithm and ddp["early_collectives_per_step"] >= 2
    assert ddp["collectives_per_step"] >= (ddp["early_collectives_per_step"] if exchange == "fp32" else 2)  # (bf16 payload: the early ranges are staged, few large collectives)
    if algorithm == "rs_ag":  # every range really went out as reduce_scatter_tensor + all_gather_into_tensor on RCCL (two collectives per range, none as a plain all_reduce)
        assert ddp["collectives_per_step"] % 2 == 0 and ddp["collectives_per_step"] >= 2 * ddp["early_collectives_per_step"], ddp
    assert ddp["payload_bytes_per_step"] > 0
    tol = 2e-3 if exchange == "bf16" else 1e-4  # (bf16 payload: the gradients are rounded once on the way through the exchange buffer)
    assert abs(rccl["config"]["final_loss"] - plain["config"]["final_loss"]) <= tol * plain["config"]["final_loss"], (rccl["config"]["final_loss"], plain["config"]["final_loss"])
