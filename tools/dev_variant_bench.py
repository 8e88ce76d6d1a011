import sys, os, json, subprocess
# usage: dev_variant_bench.py lib1.so lib2.so ... : runs bench.py (prove, small) against each library variant
for lib in sys.argv[1:]:
    env = dict(os.environ, VXPROVER_LIB=lib)
    r = subprocess.run([sys.executable, "bench.py", "--log-n", "19", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(lib, round(d["ms_per_step"], 2), {k: v for k, v in d["stage_ms_per_step"].items() if v > 0.5})
    except Exception as e:
        print(lib, "FAILED", r.stderr[-500:])
