#!/usr/bin/env python3
"""Run bench.py over a list of variants (one child process each) and print one compact line per variant.

    python tools/bench_variants.py OUTDIR "name: args ..." "name2: args ..."
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_dir = sys.argv[1]
    os.makedirs(out_dir, exist_ok=True)
    for spec in sys.argv[2:]:
        name, args = spec.split(":", 1)
        env = dict(os.environ)
        argv = args.split()
        while argv and "=" in argv[0] and not argv[0].startswith("-"):
            k, v = argv.pop(0).split("=", 1)
            env[k] = v
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=900)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not lines:
            print(f"{name}: FAILED rc={proc.returncode} {proc.stderr[-400:]!r}")
            continue
        open(os.path.join(out_dir, f"bench_{name.strip()}.json"), "w").write(lines[-1] + "\n")
        d = json.loads(lines[-1])
        ro = d["roofline"]
        parts = {k: round(v, 3) for k, v in {**ro.get("kernel_parts_ms_per_step", {}), **ro.get("other_kernels_ms_per_step", {})}.items()}
        print(f"{name.strip():28s} {d['value']:10.1f} {d['unit']:10s} ms/step {d['ms_per_step']:8.3f}  kern {ro.get('kernel_ms_per_step', 0):7.3f}  frac {ro.get('frac', 0):.4f} {ro['bound']}  "
              f"parity {(d.get('parity') or {}).get('ok')}  {parts}", flush=True)


if __name__ == "__main__":
    main()
