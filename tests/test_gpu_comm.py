"""The C ABI's own RCCL path (pbr_comm_unique_id / pbr_comm_init / pbr_allreduce_hist / pbr_halo_exchange,
include/pbr_hip.h) on ONE GPU: contract and error branches, and — with a 1-rank communicator, which RCCL allows —
the real ncclAllReduce / ncclSend / ncclRecv calls with their dlsym'd signatures.  Multi-rank behaviour is covered
by tests/test_multigpu_gloo.py (layout logic) and tests/test_gpu_tiling.py (kernels), and verified at run time by
bench.py's verification frame."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_allreduce_contract_branches(ctx):
    from direct12pbrrenderer_amd.api import PbrContext, PbrError
    hist = ctx.zeros((256,), torch.int32)
    hist[5] = 7
    ctx.allreduce_hist(hist)                      # world 1, no communicator: the local histogram is the global one
    ctx.sync()
    assert int(hist[5]) == 7 and int(hist.sum()) == 7
    c2 = PbrContext(0)
    try:
        with pytest.raises(PbrError, match="null unique id"):
            c2.comm_init(2, 0, None)
        with pytest.raises(PbrError, match="bad world/rank"):
            c2.comm_init(2, 2, b"\0" * 128)
        # a REFUSED init changes nothing (world / rank used to be overwritten before the checks): the context is still the
        # single-GPU context it was, its all-reduce the identity.  ("no communicator" is for an init that was accepted and then failed
        # inside RCCL: tests/test_gpu_comm.py::test_two_ranks_on_one_gpu_through_real_rccl checks that one where RCCL refuses.)
        hist2 = hist.clone()
        c2.allreduce_hist(hist2)
        c2.sync()
        assert torch.equal(hist2, hist)
        with pytest.raises(PbrError, match="null histogram"):
            c2.allreduce_hist(None)
    finally:
        c2.close()


_SELF_COMM = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from direct12pbrrenderer_amd.api import PbrContext, comm_unique_id
ctx = PbrContext(0)
ctx.comm_init(1, 0, comm_unique_id())           # a real 1-rank RCCL communicator
hist = torch.arange(256, dtype=torch.int32, device="cuda")
ctx.allreduce_hist(hist)                        # ncclAllReduce(sum, uint32, 256) over one rank = identity
ctx.sync()
assert torch.equal(hist.cpu(), torch.arange(256, dtype=torch.int32))
# a second init of a context that holds a communicator is refused whatever its arguments — also the world-1 / no-id form, which
# would have left world / rank at 1 / 0 with the old communicators still attached (ADVICE r05) — and changes nothing
from direct12pbrrenderer_amd.api import PbrError
for args in ((1, 0, None), (1, 0, comm_unique_id()), (2, 1, comm_unique_id())):
    try:
        ctx.comm_init(*args)
        raise SystemExit("a second pbr_comm_init was accepted: " + repr(args[:2]))
    except PbrError as e:
        assert "already has a communicator" in str(e), str(e)
ctx.allreduce_hist(hist)                        # still the 1-rank communicator
ctx.sync()
assert torch.equal(hist.cpu(), torch.arange(256, dtype=torch.int32))
# halo exchange with itself: rect A travels to rect B through pack -> ncclSend/ncclRecv -> unpack
W, H = 96, 40
plane = torch.from_numpy(np.random.default_rng(1).integers(0, 30000, (H, W, 4)).astype(np.int16)).cuda()
before = plane.cpu().numpy().copy()
peers, n = ctx.halo_peers([(0, (3, 2, 17, 9), (50, 20, 17, 9)), (0, (70, 0, 8, 40), None), (0, None, (30, 0, 8, 40))])
st = torch.zeros((ctx.halo_staging_bytes(peers, n) // 8, 4), dtype=torch.int16, device="cuda")
ctx.halo_exchange(plane.view(torch.float16), W, H, peers, n, st.view(torch.float16))
ctx.sync()
after = plane.cpu().numpy()
want = before.copy()
want[20:29, 50:67] = before[2:11, 3:20]
want[0:40, 30:38] = before[0:40, 70:78]
assert np.array_equal(after, want), "self halo exchange moved the wrong texels"
# the same exchange on the context's high-priority side stream, with work on the main stream meanwhile
before = after.copy()
other = torch.zeros(1 << 20, dtype=torch.float32, device="cuda")
ctx.side_begin()
ctx.halo_exchange(plane.view(torch.float16), W, H, peers, n, st.view(torch.float16))
ctx.side_end()
other.add_(1.0)                               # on the context's (= torch's current) stream while the strips travel
ctx.side_join()
ctx.sync()
after = plane.cpu().numpy()
want = before.copy()
want[20:29, 50:67] = before[2:11, 3:20]
want[0:40, 30:38] = before[0:40, 70:78]
assert np.array_equal(after, want) and float(other[0]) == 1.0
ctx.side_join()                               # nothing pending: a no-op
# both collectives in flight at once on different streams — the histogram all-reduce (its own communicator, an
# ncclCommSplit of the frame communicator) on the side stream, the halo exchange on the main one — and then pbr_sync
# WITHOUT a join: it waits for the context's stream and for the side-stream work not joined yet
before = after.copy()
hist = torch.arange(256, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ctx.side_begin()
ctx.allreduce_hist(hist)
ctx.side_end()
ctx.halo_exchange(plane.view(torch.float16), W, H, peers, n, st.view(torch.float16))
ctx.sync()
after = plane.cpu().numpy()
want = before.copy()
want[20:29, 50:67] = before[2:11, 3:20]
want[0:40, 30:38] = before[0:40, 70:78]
assert np.array_equal(after, want) and torch.equal(hist.cpu(), torch.arange(256, dtype=torch.int32))
ctx.side_join()
ctx.side_begin()
assert ctx.lib.pbr_ctx_use_own_stream(ctx.h) == -1 and b"side stream" in ctx.lib.pbr_last_error(ctx.h)   # refused while on the side stream
ctx.side_end()
ctx.side_join()
try:
    ctx.side_end()
    raise SystemExit("side_end without side_begin must be refused")
except Exception as e:
    assert "not on the side stream" in str(e)
ctx.close()
print("self-comm ok")
"""


@pytest.mark.timeout(300)
def test_one_rank_communicator_runs_the_rccl_calls():
    """Own process with a hard timeout: a hung collective must not take the test session (or the GPU) with it."""
    r = subprocess.run(["timeout", "-k", "10", "180", sys.executable, "-c", _SELF_COMM % ROOT], capture_output=True, text=True)
    assert r.returncode == 0 and "self-comm ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


_TWO_RANKS = r"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, %r)
rank, id_file = int(sys.argv[1]), sys.argv[2]
from direct12pbrrenderer_amd.api import PbrContext, PbrError, comm_unique_id
ctx = PbrContext(0)                              # BOTH ranks on device 0
if rank == 0:
    uid = comm_unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 60:
            raise SystemExit("rank 1: no unique id after 60 s")
        time.sleep(0.05)
    uid = open(id_file, "rb").read()
assert len(uid) == 128
try:
    ctx.comm_init(2, rank, uid)                  # ncclCommInitRank(2 ranks) + ncclCommSplit
except PbrError as e:
    # RCCL may refuse two ranks of one communicator on one device ("Duplicate GPU detected"): then the context holds NO
    # communicator at all and both collectives refuse (ADVICE r03: no half-initialised state)
    hist = torch.arange(256, dtype=torch.int32, device="cuda")
    for call in (lambda: ctx.allreduce_hist(hist),
                 lambda: ctx.halo_exchange(torch.zeros((4, 4, 4), dtype=torch.float16, device="cuda"), 4, 4, *ctx.halo_peers([(1 - rank, (0, 0, 1, 1), (2, 2, 1, 1))]),
                                           torch.zeros((2, 4), dtype=torch.float16, device="cuda"))):
        try:
            call()
            raise SystemExit("a collective ran although pbr_comm_init had failed")
        except PbrError as e2:
            assert "no communicator" in str(e2), e2
    print("REFUSED:", e)
    ctx.close()
    raise SystemExit(77)
# ---- histogram all-reduce on the split communicator: rank r contributes (r + 1) * bin
hist = (torch.arange(256, dtype=torch.int32) * (rank + 1)).cuda()
ctx.allreduce_hist(hist)
ctx.sync()
assert torch.equal(hist.cpu(), torch.arange(256, dtype=torch.int32) * 3), "ncclAllReduce(sum, uint32) over two ranks"
# ---- halo exchange between the two ranks: each sends its rectangle A, receives the other's into B
W, H = 96, 40
def plane_of(r):
    return np.random.default_rng(100 + r).integers(0, 30000, (H, W, 4)).astype(np.int16)
mine, theirs = plane_of(rank), plane_of(1 - rank)
plane = torch.from_numpy(mine).cuda()
send = {0: (3, 2, 17, 9), 1: (40, 11, 17, 9)}
recv = (60, 25, 17, 9)
peers, n = ctx.halo_peers([(1 - rank, send[rank], recv)])
st = torch.zeros((ctx.halo_staging_bytes(peers, n) // 8, 4), dtype=torch.int16, device="cuda")
def expect(before):
    want = before.copy()
    sx, sy, sw, sh = send[1 - rank]
    want[recv[1]:recv[1] + recv[3], recv[0]:recv[0] + recv[2]] = theirs[sy:sy + sh, sx:sx + sw]
    return want
ctx.halo_exchange(plane.view(torch.float16), W, H, peers, n, st.view(torch.float16))
ctx.sync()
assert np.array_equal(plane.cpu().numpy(), expect(mine)), "ncclSend / ncclRecv between two ranks moved the wrong texels"
# ---- both communicators in flight at once: all-reduce on the side stream, halo exchange on the main one
plane.copy_(torch.from_numpy(mine).cuda())
hist = (torch.arange(256, dtype=torch.int32) * (rank + 1)).cuda()
torch.cuda.synchronize()
ctx.side_begin()
ctx.allreduce_hist(hist)
ctx.side_end()
ctx.halo_exchange(plane.view(torch.float16), W, H, peers, n, st.view(torch.float16))
ctx.side_join()
ctx.sync()
assert torch.equal(hist.cpu(), torch.arange(256, dtype=torch.int32) * 3)
assert np.array_equal(plane.cpu().numpy(), expect(mine))
ctx.close()
print("two-rank ok")
"""


@pytest.mark.timeout(400)
def test_two_ranks_on_one_gpu_through_real_rccl(tmp_path):
    """What no one-rank communicator can show: ncclSend / ncclRecv between DISTINCT ranks and the
    ncclCommSplit communicator's all-reduce under concurrency, through the C ABI.  Two fresh child processes, both on
    device 0, one 2-rank RCCL communicator; hard timeouts.  If RCCL on this pool refuses two ranks per device the test
    skips with RCCL's reason (and has then checked that the context whose pbr_comm_init failed inside RCCL keeps the world it was
    asked for WITHOUT a communicator: both collectives refuse — it is never half in multi-GPU mode, and never silently single-GPU)."""
    id_file = str(tmp_path / "rccl_unique_id")
    env = dict(os.environ, NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen(["timeout", "-k", "10", "150", sys.executable, "-c", _TWO_RANKS % ROOT, str(r), id_file],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in (0, 1)]
    outs = [p.communicate(timeout=200) for p in procs]
    codes = [p.returncode for p in procs]
    if 77 in codes or any("REFUSED:" in o[0] for o in outs):
        reason = next((line for o in outs for line in o[0].splitlines() if line.startswith("REFUSED:")), "REFUSED")
        warn = next((line for o in outs for line in o[1].splitlines() if "WARN" in line or "uplicate" in line), "")
        pytest.skip(f"RCCL refuses two ranks of one communicator on one device here: {reason} {warn}".strip())
    assert codes == [0, 0] and all("two-rank ok" in o[0] for o in outs), (codes, [o[0][-1500:] for o in outs], [o[1][-3000:] for o in outs])


_TAIL_OVERLAP = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from direct12pbrrenderer_amd import scene, synth
from direct12pbrrenderer_amd.api import PbrContext, comm_unique_id
from direct12pbrrenderer_amd.pipeline import DeferredFrame, TileSpec
import bench
ctx = PbrContext(0)
ctx.comm_init(1, 0, comm_unique_id())            # a real 1-rank RCCL communicator: the all-reduce is the C ABI's
lut, env, sh = bench.build_ibl(ctx)
W, H = 1280, 720
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh, delta_time=1.0 / 60.0)
lights = synth.lights_in_view_box(64, cam)
gbs = [synth.gbuffer_tile(0, 0, W, H, W, H), synth.gbuffer_tile(0, 0, W, H, W, H, rough_min=96)]
def run(overlap, allreduce=True, from_bloom=False):
    fr = DeferredFrame(ctx, TileSpec(0, 0, W, H, W, H, 0), g, lights, lut, 512, env, 512, 5, allreduce=ctx.allreduce_hist if allreduce else None)
    fr.set_prev_luminance(0.18)
    if overlap:
        fr.enable_tail_overlap(from_bloom=from_bloom)
    out = []
    for i in range(7):
        fr.upload_gbuffer(gbs[i %% 2])               # alternating inputs: a stale buffer would show
        fr.render()
        if overlap and i %% 2:
            fr.finish()
        ctx.sync()                                   # also waits for the side stream's un-joined tail (even frames rely on it)
        out.append((fr.ldr.cpu().numpy().copy(), float(fr.avg.cpu()[0])))
    # and back to back without a host synchronisation in between: the last frame must still be the same
    for i in range(7, 12):
        fr.upload_gbuffer(gbs[i %% 2])
        fr.render()
    fr.finish(); ctx.sync()
    out.append((fr.ldr.cpu().numpy().copy(), float(fr.avg.cpu()[0])))
    return out
try:
    fr = DeferredFrame(ctx, TileSpec(0, 0, W, H, W, H, 0), g, lights, lut, 512, env, 512, 5, allreduce=lambda h: None)
    fr.enable_tail_overlap()
    raise SystemExit("an all-reduce that is not the C ABI's must be refused")
except ValueError as e:
    assert "C ABI" in str(e)
a = run(False)
# with the RCCL all-reduce in the tail (multi-GPU frames); without one (N = 1): the tail alone, and everything behind the shade
for what, b in (("tail + all-reduce", run(True)), ("tail, no all-reduce", run(True, allreduce=False)),
                ("bloom + tail, no all-reduce", run(True, allreduce=False, from_bloom=True)), ("bloom + tail + all-reduce", run(True, from_bloom=True))):
    for i, ((la, va), (lb, vb)) in enumerate(zip(a, b)):
        assert va == vb, (what, i, va, vb)
        assert np.array_equal(la, lb), (what, i, int((la != lb).sum()))
ctx.close()
print("tail overlap ok")
"""


@pytest.mark.timeout(400)
def test_overlapped_frame_tail_equals_the_plain_order():
    """DeferredFrame.enable_tail_overlap (what bench.py's multi-GPU frames run when the histogram all-reduce is the C ABI's, and —
    round 4 — its N = 1 frames for the `*_tail_overlapped` figures): all-reduce + average + tone-map of frame i (from_bloom: the
    bloom chain too) on the context's side stream beside frame i + 1's shade, HDR target and histogram double-buffered.  With a 1-rank RCCL communicator on the one GPU: LDR image and adapted luminance of every
    frame of an alternating input sequence identical to the plain order, synchronised frame by frame and back to back."""
    r = subprocess.run(["timeout", "-k", "10", "300", sys.executable, "-c", _TAIL_OVERLAP % ROOT], capture_output=True, text=True)
    assert r.returncode == 0 and "tail overlap ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.timeout(400)
def test_cu_partition_renders_the_same_frames():
    """pbr_ctx_set_cu_masks (knobs build only since round 6: the subprocesses load libpbr_hip_knobs.so): the device's CUs split between
    the context's private stream and its side stream (64 CUs = 8 per XCD for the side stream) — in order and in post-shade throughput mode the frames are those of the unpartitioned in-order render; and the
    contract: an empty mask, or a call while side work is pending, is refused.  Own process: the masks recreate the context's streams."""
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from direct12pbrrenderer_amd.api import PbrContext, PbrError
ctx = PbrContext(0)
ctx.use_own_stream()
zero = np.zeros(8, np.uint32); full = np.full(8, 0xFFFFFFFF, np.uint32)
for args, why in (((zero.ctypes.data, full.ctypes.data, 8), "empty CU mask"), ((full.ctypes.data, full.ctypes.data, 0), "mask words")):
    try:
        ctx._check(ctx.lib.pbr_ctx_set_cu_masks(ctx.h, *args)); raise SystemExit("accepted: " + why)
    except PbrError as e:
        assert why in str(e), e
ctx.side_begin()
try:
    ctx.partition_cus(64); raise SystemExit("accepted while on the side stream")
except PbrError as e:
    assert "pending" in str(e), e
ctx.side_end(); ctx.side_join(); ctx.sync()
ctx.close()
print("contract ok")
""" % ROOT
    env = dict(os.environ, PBR_HIP_LIB=os.path.join(ROOT, "direct12pbrrenderer_amd", "libpbr_hip_knobs.so"))
    r = subprocess.run(["timeout", "-k", "10", "120", sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "contract ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    # the product library refuses with a pointer to the knobs build
    r = subprocess.run(["timeout", "-k", "10", "120", sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nfrom direct12pbrrenderer_amd.api import PbrContext, PbrError\n"
                        "c = PbrContext(0)\ntry:\n    c.partition_cus(64); print('accepted')\nexcept PbrError as e:\n    print('refused:', e)" % ROOT], capture_output=True, text=True)
    assert r.returncode == 0 and "refused:" in r.stdout and "knobs build" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
    r = subprocess.run(["timeout", "-k", "10", "300", sys.executable, os.path.join(ROOT, "tools", "cu_partition.py"), "low", "0", "64"], capture_output=True, text=True, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("[low] side CUs")]
    assert r.returncode == 0 and len(lines) == 2 and all(l.endswith("True") for l in lines), (r.returncode, r.stdout[-1500:], r.stderr[-3000:])


def test_halo_pack_unpack_roundtrip(ctx):
    """The two halves of pbr_halo_exchange on their own (what a non-RCCL transport uses): staging layout = all send
    rectangles in peer order, then all recv rectangles."""
    W, H = 80, 24
    rng = np.random.default_rng(3)
    plane = torch.from_numpy(rng.integers(0, 30000, (H, W, 4)).astype(np.int16)).cuda()
    before = plane.cpu().numpy().copy()
    plan = [(1, (0, 0, 5, 24), (10, 0, 5, 24)), (2, (20, 4, 30, 3), (40, 20, 30, 3))]
    peers, n = ctx.halo_peers(plan)
    nb = ctx.halo_staging_bytes(peers, n)
    assert nb == 8 * 2 * (5 * 24 + 30 * 3)
    st = torch.zeros((nb // 8, 4), dtype=torch.int16, device="cuda")
    ctx.halo_pack(plane.view(torch.float16), W, H, peers, n, st.view(torch.float16), unpack=False)
    ctx.sync()
    s = st.cpu().numpy()
    assert np.array_equal(s[:120].reshape(24, 5, 4), before[0:24, 0:5]) and np.array_equal(s[120:210].reshape(3, 30, 4), before[4:7, 20:50])
    s2 = st.clone()
    s2[210:330] = st[:120]          # loop the first send rectangle back as the first recv rectangle ...
    s2[330:420] = st[120:210]       # ... and the second
    ctx.halo_pack(plane.view(torch.float16), W, H, peers, n, s2.view(torch.float16), unpack=True)
    ctx.sync()
    after = plane.cpu().numpy()
    want = before.copy()
    want[0:24, 10:15] = before[0:24, 0:5]
    want[20:23, 40:70] = before[4:7, 20:50]
    assert np.array_equal(after, want)
    from direct12pbrrenderer_amd.api import PbrError
    with pytest.raises(PbrError, match="outside the plane"):
        bad, nbad = ctx.halo_peers([(1, (70, 0, 20, 4), None)])
        ctx.halo_pack(plane.view(torch.float16), W, H, bad, nbad, st.view(torch.float16))


def test_membench_read_touches_every_byte(ctx):
    """The HBM-read probe of bench.py really reads the buffer: the xor of all words comes out right."""
    n = 1 << 20
    rng = np.random.default_rng(9)
    a = rng.integers(0, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    buf = torch.from_numpy(a).cuda()
    blocks = 64
    sink = torch.zeros(blocks, dtype=torch.int32, device="cuda")
    ctx.membench_read(buf, sink, blocks)
    ctx.sync()
    got = np.bitwise_xor.reduce(sink.cpu().numpy().view(np.uint32))
    assert got == np.bitwise_xor.reduce(a.view(np.uint32))


def test_valubench_stamps_cycles_and_clock(ctx):
    """The VALU issue-rate probe of bench.py: every wave reports more shader cycles than instructions it issued (a wave64 fp32
    instruction cannot issue faster than one per cycle), a clock between 0.5 and 3 GHz from the two counters, and the slow classes
    cost more cycles than the plain one; bad arguments are refused."""
    # five waves per SIMD (the shade's occupancy): there the three issue classes are well apart (per wave ~7.7 / 15 / 24 cycles per
    # instruction on this part; with one wave per SIMD they are 7.5-8 / 9 / 11.5 and an ordering assert would sit on the noise)
    blocks, iters = 5 * torch.cuda.get_device_properties(0).multi_processor_count, 2000
    st = torch.zeros((blocks * 4, 4), dtype=torch.int64, device="cuda")
    per_inst = {}
    for op in (0, 2, 3):
        st.zero_()
        ctx.valubench(op, blocks, iters, st)
        ctx.sync()
        s = st.cpu().numpy()
        cyc, ticks = s[:, 1] - s[:, 0], s[:, 3] - s[:, 2]
        assert (cyc > iters * 8).all() and (ticks > 0).all()
        clock = np.median(cyc / ticks) * 100e6
        assert 0.5e9 < clock < 3.0e9, clock
        per_inst[op] = float(np.median(cyc)) / (iters * 8)
    assert 1.2 * per_inst[0] < per_inst[2] and 1.2 * per_inst[2] < per_inst[3], per_inst
    from direct12pbrrenderer_amd.api import PbrError
    with pytest.raises(PbrError):
        ctx.valubench(7, blocks, iters, st)
