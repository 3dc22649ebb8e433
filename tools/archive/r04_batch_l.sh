# round-4 batch l: the shade with a block's pixels sorted by list length (knobs build, PBR_SHADE_SORT=1): same image? how fast?
mkdir -p gpurun_out
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_l_sort_check.txt
import os, subprocess, sys
code = r"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from direct12pbrrenderer_amd import scene, synth
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
out = {}
for (W, H) in ((3840, 2160), (1000, 531)):
    cam = scene.Camera.reference_default(W, H)
    g = scene.make_global(cam, W, H, sh_pack=sh)
    fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H) if (W, H) == (3840, 2160) else __import__('direct12pbrrenderer_amd.pipeline', fromlist=['TileSpec']).TileSpec(0, 0, W, H, W, H, 0), g, synth.lights_in_view_box(256, cam), lut, 512, env, 512, 5)
    fr.upload_gbuffer(synth.gbuffer_tile(0, 0, W, H, W, H, coverage_mask=True))
    fr.clustered(); fr.shade(); ctx.sync()
    out[(W, H)] = fr.hdr.cpu().view(torch.int16).numpy().copy()
np.savez(sys.argv[1], a=out[(3840, 2160)], b=out[(1000, 531)])
"""
for tag, env in (("plain", {}), ("sorted", {"PBR_SHADE_SORT": "1"})):
    subprocess.check_call([sys.executable, "-c", code, f"/tmp/shade_{tag}.npz"], env=dict(os.environ, **env))
import numpy as np
p, s = np.load("/tmp/shade_plain.npz"), np.load("/tmp/shade_sorted.npz")
for k in ("a", "b"):
    print(k, "identical" if np.array_equal(p[k], s[k]) else f"DIFFERENT in {int((p[k] != s[k]).sum())} values")
PY
run() { tag=$1; shift
  env "$@" python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-graph --no-shade-paths --no-tail-overlap > gpurun_out/ab_$tag.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/ab_$tag.json'));s=d['roofline']['stage_ms'];print('$tag', 'frame', d['ms_per_step'], 'shade in-frame', s['shade(in frame)'])" || exit 1
}
for r in 1 2; do run plain X=1; run sorted PBR_SHADE_SORT=1; done 2>&1 | tee gpurun_out/r04_l_ab_shade_sort.txt
