"""Diagnostic: section clocks of the trace megakernel (librt_hip_stamps.so, built with -DRT_STAMPS)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuraytracer_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "cpuraytracer_amd", "lib", "librt_hip_stamps.so")
from cpuraytracer_amd import HipRenderer, scenes
r = HipRenderer(0)
SCENE = sys.argv[1] if len(sys.argv) > 1 else "cover"
r.upload(scenes.build_scene(SCENE, 1, 1200, 800))
L = _capi.load()
out = (C.c_ulonglong * 20)()
L.rt_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
r.render(1200, 800, 1, 17, 50, 1)
L.rt_debug_stamps(r._h, out)
st = r.render(1200, 800, 1, 33, 50, 1)
L.rt_debug_stamps(r._h, out)
v = list(out)
tot = v[0] + v[1] + v[2]
print(json.dumps({"wave_iterations": v[3], "cycles_per_iteration": tot / v[3], "share_refill": v[0] / tot, "share_scan": v[1] / tot,
                  "share_transitions_hit": v[2] / tot, "scan_filter_share": v[4] / (v[4] + v[5]), "scan_resolve_share": v[5] / (v[4] + v[5]),
                  "resolve_items_per_ray_scan": v[6] / st.traversals, "resolve_max_items_per_wave_iteration": v[7] / v[3],
                  "lanes_live_per_iteration": st.traversals / v[3],
                  "phaseA_cycles_per_iteration": v[8] / v[3], "phaseB_cycles_per_iteration": v[9] / v[3],
                  "phaseA_steps_per_iteration": v[10] / v[3], "phaseB_steps_per_iteration": v[11] / v[3],
                  "hit_scatter_cycles": v[12] / v[3], "hit_shadow_query_cycles": v[13] / v[3], "hit_shade_cycles": v[14] / v[3], "lane0_hit_iterations": v[17], "hit_pre_cycles_when_lane0_hits": v[15] / max(1, v[17]), "hit_post_cycles_when_lane0_hits": v[16] / max(1, v[17]), "ms_trace_stamped_build": st.ms_render}))
