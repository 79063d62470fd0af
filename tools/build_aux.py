"""Auxiliary builds of libspath_hip.so under build/ (git-ignored; they travel to the GPU box with the snapshot):
  libspath_hip_stats.so     -DSP_FILTER_STATS   survivor / round / exact-turn statistics on stderr of sphip_get_stats (tools/collect_profiles.sh)
  libspath_hip_all.so       -DSP_ALL_VARIANTS   every scan generation (tools/soak.py, tools/pytest_with_lib.py)
  libspath_hip_unpinned.so  -DSP_CYLM_UNPINNED  the float->half conversion defect switched back on (tests/test_hip_stage1_audit.py builds it itself too)
  phase_timers.so           -DSP_PHASE_TIMERS   wave lifetime by phase (tools/phase_timers.py)
python tools/build_aux.py [stats all unpinned timers]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
W = {"stats": ("libspath_hip_stats.so", ["-DSP_FILTER_STATS"]), "all": ("libspath_hip_all.so", ["-DSP_ALL_VARIANTS"]),
     "unpinned": ("libspath_hip_unpinned.so", ["-DSP_CYLM_UNPINNED"]), "timers": ("phase_timers.so", ["-DSP_PHASE_TIMERS"])}
os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
procs = []
for k in (sys.argv[1:] or list(W)):
    out, fl = W[k]
    procs.append((k, subprocess.Popen(g.hipcc_command(os.path.join(ROOT, "build", out), fl), cwd=ROOT)))
for k, p in procs:
    if p.wait():
        sys.exit(f"build of {k} failed")
    print("built", W[k][0])
