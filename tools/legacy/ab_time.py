"""A/B timing of alternative builds (build/abl_*.so) on the configs[2] frame: python tools/ab_time.py [spp] [variants...]"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spp = sys.argv[1] if len(sys.argv) > 1 else "32"
variants = sys.argv[2:] or ["rpl_cyl2s", "rpl_cyl4s"]
for lib in sorted(glob.glob(os.path.join(root, "build", "abl_*.so"))):
    for v in variants:
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "render_once.py"), v, "1920", "1080", spp, "10000", lib],
                           capture_output=True, text=True, env=dict(os.environ, REPS="2"))
        lines = [l for l in p.stdout.splitlines() if l.startswith(v)]
        print(f"{os.path.basename(lib):18s} {lines[-1] if lines else 'FAILED ' + p.stderr[-300:]}", flush=True)
