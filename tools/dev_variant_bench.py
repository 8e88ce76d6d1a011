import sys, os, json, subprocess
# usage: dev_variant_bench.py [--log-n N] lib1.so lib2.so ... : runs bench.py (prove) against each library variant (VXPROVER_LIB)
args = sys.argv[1:]
log_n = "21"
if args and args[0] == "--log-n":
    log_n, args = args[1], args[2:]
for lib in args:
    env = dict(os.environ, VXPROVER_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "bench.py", "--log-n", log_n, "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-host-witness-leg"],
                       capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(lib, round(d["ms_per_step"], 2), {k: v for k, v in d["stage_ms_per_step"].items() if v > 0.5}, flush=True)
    except Exception as e:
        print(lib, "FAILED", r.stderr[-500:], flush=True)
