"""Frame time of the C++ pass graph (libpbr_host.so: RenderScheduler -> FrameGraph -> passes -> HipCommandList -> C ABI) at
4K with 256 lights: every reference dispatch issued one by one / fused passes, each with the reference's per-frame fence
wait (D3D12Device::EndFrame) and in throughput mode (3 frames in flight).  The bench line carries the same figures in its
`host_graph` block; this tool is what the rocprofv3 kernel trace of the drop-in path runs (tools/host_trace.sh).
usage: python tools/host_frame_ms.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.host import HostRenderer  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H, ENV, LUT = 3840, 2160, 512, 512
cam = scene.Camera.reference_default(W, H)
r = HostRenderer(0, W, H, ENV, LUT)
r.set_skybox(synth.env_cube(ENV), ENV)
r.set_lights(synth.lights_in_view_box(256, cam))
r.set_gbuffer(synth.gbuffer_tile(0, 0, W, H, W, H))
r.set_initial_luminance(0.18)
r.render()
for fused in (0, 1):
    r.set_fused(fused)
    for in_flight in (1, 3):
        r.set_frames_in_flight(in_flight)
        r.render_n(frames)
        ms = r.render_n(frames)
        print(f"host pass graph, {'fused passes' if fused else 'dispatch by dispatch'}, {'fence per frame' if in_flight == 1 else '3 frames in flight'}: "
              f"{ms:.4f} ms/frame ({r.dispatch_count()} dispatches/frame) = {W * H / ms / 1e3:.0f} Mpixel/s", flush=True)
# the first frame's IBL precompute through the pass API (PreFilterEnvMapPass::Execute, DeferredPipeline.cpp:77-115): a frame right
# after set_skybox (which invalidates the pass) minus a steady frame, dispatch by dispatch (five pbr_prefilter_env_mip) and fused
# (one pbr_prefilter_env)
r.set_frames_in_flight(1)
for fused in (0, 1):
    r.set_fused(fused)
    steady = r.render_n(20)
    first = []
    for _ in range(3):
        r.set_skybox(synth.env_cube(ENV), ENV)
        first.append(r.render_n(1))
    print(f"host pass graph, {'fused passes' if fused else 'dispatch by dispatch'}: frame with the env prefilter {min(first):.3f} ms, steady frame {steady:.3f} ms "
          f"-> PreFilterEnvMapPass {min(first) - steady:.3f} ms", flush=True)
r.set_fused(1)
r.set_frames_in_flight(3)
r.set_tail_overlap(True)
r.render_n(frames)
ms = r.render_n(frames)
print(f"host pass graph, fused passes, 3 frames in flight, frame tail on the side stream: {ms:.4f} ms/frame = {W * H / ms / 1e3:.0f} Mpixel/s", flush=True)
r.close()
