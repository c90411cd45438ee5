"""Worker bodies of the multi-process GPU tests: several ranks share ONE MI355X (the test boxes
have one GPU), which RCCL refuses ("duplicate GPU") - so the ranks run the native engine's device
plane without a communicator (DDL_RCCL_SKIP_COMM=1) and exchange gradients through the direct
xGMI all-reduce kernel over REAL hipIpc mappings between REAL processes."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import distributeddeeplearning_amd.torch as hvd  # noqa: E402


_T0 = time.perf_counter()


def stamp(what):
    if os.environ.get("TEST_STAMPS") == "1":
        print("STAMP rank %s %8.3f s  %s" % (os.environ.get("RANK"), time.perf_counter() - _T0, what), flush=True)


def emit(**kw):
    print("RESULT " + json.dumps(kw, sort_keys=True), flush=True)


def _busy(ms):
    """Keep the device busy for roughly `ms` on the current stream (uneven load between ranks)."""
    x = torch.randn(2048, 2048, device="cuda")
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        x = (x @ x).tanh_()
    return x


def scenario_xgmi_words():
    """Every word of 3 epochs x {two-shot, one-shot} x {bf16, fp32} x several sizes checked, with
    rank 1 delayed on the device (its stream is busy) and on the host."""
    hvd.init()
    r, n = hvd.rank(), hvd.size()
    comm = hvd.comm()
    assert hvd.engine_name() == "native" and comm.algo != "rccl"
    bad = []
    arena = {dt: torch.zeros(3 << 20, dtype=dt, device="cuda") for dt in (torch.bfloat16, torch.float32)}
    for dt in (torch.bfloat16, torch.float32):
        comm.register_region(arena[dt])
    launches = 0
    counts0 = hvd.comm().launch_counts()   # the init self-test already ran checked all-reduces through the kernel
    assert counts0["xgmi"] >= 2, counts0   # (bf16 + fp32 on the enabled algorithm)
    stamp("regions registered")
    for epoch in range(int(os.environ.get("TEST_EPOCHS", "3"))):
        stamp("epoch %d" % epoch)
        for dt in (torch.bfloat16, torch.float32):
            for numel, off in ((16 * n, 0), (4096 * n, 1024), (1 << 20, 65536), ((3 << 20) - 16 * n * 8, 16 * n * 8)):
                numel = numel // (8 * n) * (8 * n)
                buf = arena[dt][off: off + numel]
                base = torch.arange(numel, device="cuda", dtype=torch.float32) % 251
                val = base * (r + 1) + epoch            # small integers: exact in bf16 up to 256*... keep < 2^8 steps
                buf.copy_(val.to(dt))
                if r == 1 and os.environ.get("TEST_NO_DELAY") != "1":
                    _busy(15 if epoch == 1 else 2)
                    time.sleep(0.01 * epoch)
                for algo, op in (("twoshot", "average"), ("oneshot", "sum")):
                    comm.algo = algo
                    buf.copy_(val.to(dt))
                    # expected: per-rank values rounded to dt, summed in fp32, scaled, rounded once
                    ranks = [(base * (q + 1) + epoch).to(dt).float() for q in range(n)]
                    exp = sum(ranks[1:], ranks[0])
                    if op == "average":
                        exp = exp * (1.0 / n)
                    exp = exp.to(dt)
                    stamp("collective %d %s %s %d" % (launches, algo, str(dt)[6:], numel))
                    h = hvd.allreduce_async_(buf, op=hvd.Average if op == "average" else hvd.Sum,
                                             name="w.%s.%s.%d.%d" % (algo, str(dt)[6:], numel, epoch))
                    hvd.synchronize(h)
                    launches += 1
                    torch.cuda.synchronize()
                    if not torch.equal(buf, exp):
                        bad.append((epoch, str(dt), numel, algo, int((buf != exp).sum())))
    counts = comm.launch_counts()
    time.sleep(0.5)
    st = comm.stats()
    ms = sorted(v["mean_ms"] for v in st.values() if v["count"])
    emit(rank=r, bad=bad, launches=launches, xgmi=counts["xgmi"] - counts0["xgmi"], rccl=counts["rccl"] - counts0["rccl"],
         failed=comm.failed(), stats=len(st), ms_min=ms[0] if ms else -1, ms_median=ms[len(ms) // 2] if ms else -1,
         ms_max=ms[-1] if ms else -1)
    hvd.shutdown()


def scenario_xgmi_timeout():
    """Rank 1 skips a collective: rank 0's kernel must give up (bounded spins), the watchdog must
    fail the engine and the next wait must raise in Python - nothing may hang."""
    hvd.init()
    r = hvd.rank()
    comm = hvd.comm()
    buf = torch.ones(1 << 16, dtype=torch.bfloat16, device="cuda")
    comm.register_region(buf)
    hvd.synchronize(hvd.allreduce_async_(buf, average=True, name="ok"))
    torch.cuda.synchronize()
    raised, kind = False, ""
    t0 = time.perf_counter()
    if r == 0:
        try:
            h = hvd.allreduce_async_(buf, average=True, name="lonely")
            hvd.synchronize(h)
            torch.cuda.synchronize()
            time.sleep(1.0)   # the watchdog notices the kernel's timeout word within its period
            hvd.synchronize(hvd.allreduce_async_(buf, average=True, name="after"))
        except Exception as e:  # noqa: BLE001
            raised, kind = True, type(e).__name__ + ": " + str(e)[:160]
    else:
        time.sleep(3.0)
    emit(rank=r, raised=raised, kind=kind, seconds=round(time.perf_counter() - t0, 2), failed=comm.failed())
    os._exit(0)


def scenario_dp_equivalence():
    """2 ranks x batch 4 through the HIP bf16 path and the xGMI kernel == the average of the two
    ranks' gradients computed in ONE process (bit for bit: the kernel sums in fp32 and rounds
    once, the weight-gradient reduction is ordered), and the replicas stay identical."""
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD

    hvd.init()
    r, n = hvd.rank(), hvd.size()
    dev = torch.device("cuda", 0)

    def make():
        torch.manual_seed(7)
        m = resnet18(num_classes=64).to_compute(dev, torch.bfloat16).train()
        return m

    g = torch.Generator().manual_seed(5)
    xs = torch.rand(n, 4, 3, 64, 64, generator=g).to(dev).to(torch.bfloat16)
    xs = [xs[i].contiguous(memory_format=torch.channels_last) for i in range(n)]
    ys = [torch.randint(0, 64, (4,), generator=g).to(dev) for _ in range(n)]

    # reference: this process computes EVERY rank's gradient on a private copy of the model
    ref_grads = []
    for q in range(n):
        m = make()
        o = FusedSGD(m.named_parameters(), lr=0.05, momentum=0.9)
        o.zero_grad()
        ops.softmax_cross_entropy(m(xs[q]), ys[q]).backward()
        torch.cuda.synchronize()
        ref_grads.append({dt: t.clone() for dt, t in o.arena.grads.items()})
        del m, o
    from distributeddeeplearning_amd.ops import hip_ops
    hip_ops.reset_caches()
    stamp("reference passes done")

    model = make()
    if r != 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    hvd.broadcast_parameters(model.state_dict(), root_rank=0)
    stamp("parameters broadcast")
    opt = hvd.DistributedOptimizer(FusedSGD(model.named_parameters(), lr=0.05, momentum=0.9),
                                   named_parameters=model.named_parameters())
    stamp("optimizer wrapped (regions registered)")
    opt.zero_grad()
    loss = ops.softmax_cross_entropy(model(xs[r]), ys[r])
    stamp("forward issued")
    loss.backward()
    stamp("backward issued")
    opt.synchronize()
    stamp("synchronize() returned")
    torch.cuda.synchronize()
    stamp("device idle")
    mism, detail = {}, {}
    for dt, got in opt.arena.grads.items():
        acc = ref_grads[0][dt].float()
        for q in range(1, n):
            acc = acc + ref_grads[q][dt].float()
        exp = (acc * (1.0 / n)).to(got.dtype)
        bad = got != exp
        mism[str(dt)] = int(bad.sum())
        if mism[str(dt)]:   # post-mortem: where, and what do the wrong words hold?
            per_bucket = {b.index: int(bad[b.start:b.end].sum()) for b in opt.arena.plans[dt].buckets}
            same_as = {"rank%d_local" % q: int((bad & (got == ref_grads[q][dt])).sum()) for q in range(n)}
            idx = bad.nonzero().flatten()
            detail[str(dt)] = {"per_bucket": per_bucket, "equals": same_as, "first": int(idx[0]), "last": int(idx[-1]),
                               "max_abs_err": float((got.float() - exp.float()).abs().max()),
                               "max_rel_to_absmax": float((got.float() - exp.float()).abs().max() / exp.float().abs().max())}
    opt._ddl_reset_buckets()
    # two full steps: replicas must stay bit-identical
    trace = []
    mode = os.environ.get("TEST_STEP_MODE", "step")
    for it in range(2):
        opt.zero_grad()
        ops.softmax_cross_entropy(model(xs[r]), ys[r]).backward()
        if mode == "sync":
            opt.synchronize()
            torch.cuda.synchronize()
            trace.append(["grads", it] + [float(g.double().sum()) for g in opt.arena.grads.values()])
        opt.step()
        torch.cuda.synchronize()
        trace.append(["params", it] + [float(w.double().sum()) for w in opt.arena.params.values()])
        trace.append(["moment", it] + [float(v.double().sum()) for v in opt.momentum_buf.values()])
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
    counts = hvd.comm().launch_counts()
    emit(rank=r, mismatches=mism, checksum=float(flat.double().sum()), absmax=float(flat.abs().max()),
         xgmi=counts["xgmi"], metrics=opt.comm_metrics(), loss=float(loss), detail=detail, trace=trace)
    hvd.shutdown()


def scenario_resume():
    """Train 2 steps data-parallel, rank 0 saves; fresh, DIFFERENT models on both ranks resume."""
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD
    from distributeddeeplearning_amd.utils import checkpoint

    hvd.init()
    r = hvd.rank()
    dev = torch.device("cuda", 0)
    d = os.environ["TEST_CKPT_DIR"]

    def make(seed):
        torch.manual_seed(seed)
        m = resnet18(num_classes=32).to_compute(dev, torch.bfloat16).train()
        o = hvd.DistributedOptimizer(FusedSGD(m.named_parameters(), lr=0.05, momentum=0.9),
                                     named_parameters=m.named_parameters())
        return m, o

    def sums(m, o, buffers=True):
        torch.cuda.synchronize()
        vals = [v.double().sum().item() for v in o.master.values()] + \
               [v.double().sum().item() for v in o.momentum_buf.values()] + \
               [v.double().sum().item() for v in o.arena.params.values()]
        if buffers:   # BatchNorm statistics are per-rank by design (no SyncBatchNorm in the reference)
            vals += [b.double().sum().item() for _, b in sorted(m.named_buffers())]
        return [float(v) for v in vals]

    g = torch.Generator().manual_seed(21 + r)
    x = torch.rand(4, 3, 64, 64, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 32, (4,), generator=g).to(dev)

    def step(m, o):
        o.zero_grad()
        ops.softmax_cross_entropy(m(x), y).backward()
        o.step()

    m, o = make(7)
    hvd.broadcast_parameters(m.state_dict(), root_rank=0)
    o.refresh_master_from_params()
    for _ in range(2):
        step(m, o)
    saved = sums(m, o)
    checkpoint.save(d, 2, m, o)          # rank 0 writes, the others return None
    hvd.barrier()
    stamp("saved")
    del m, o
    m2, o2 = make(100 + r)               # different on every rank
    epoch = checkpoint.resume(d, m2, o2)
    after = sums(m2, o2)
    stamp("resumed")
    step(m2, o2)
    emit(rank=r, epoch=epoch, saved=saved if r == 0 else None, after_resume=after, after_step=sums(m2, o2, buffers=False),
         files=sorted(f for f in os.listdir(d) if f.endswith(".pt")))
    hvd.shutdown()


if __name__ == "__main__":
    globals()["scenario_" + sys.argv[1]]()
