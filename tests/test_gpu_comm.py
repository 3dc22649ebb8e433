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
        with pytest.raises(PbrError, match="no communicator"):   # world was recorded as 2, the init never completed
            c2.allreduce_hist(hist)
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
def run(overlap):
    fr = DeferredFrame(ctx, TileSpec(0, 0, W, H, W, H, 0), g, lights, lut, 512, env, 512, 5, allreduce=ctx.allreduce_hist)
    fr.set_prev_luminance(0.18)
    if overlap:
        fr.enable_tail_overlap()
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
a, b = run(False), run(True)
for i, ((la, va), (lb, vb)) in enumerate(zip(a, b)):
    assert va == vb, (i, va, vb)
    assert np.array_equal(la, lb), (i, int((la != lb).sum()))
ctx.close()
print("tail overlap ok")
"""


@pytest.mark.timeout(400)
def test_overlapped_frame_tail_equals_the_plain_order():
    """DeferredFrame.enable_tail_overlap (what bench.py's multi-GPU frames run when the histogram all-reduce is the C ABI's):
    all-reduce + average + tone-map of frame i on the context's side stream beside frame i + 1's shade, HDR target and
    histogram double-buffered.  With a 1-rank RCCL communicator on the one GPU: LDR image and adapted luminance of every
    frame of an alternating input sequence identical to the plain order, synchronised frame by frame and back to back."""
    r = subprocess.run(["timeout", "-k", "10", "300", sys.executable, "-c", _TAIL_OVERLAP % ROOT], capture_output=True, text=True)
    assert r.returncode == 0 and "tail overlap ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


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
