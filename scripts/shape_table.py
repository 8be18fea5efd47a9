"""Runs bench.py over the BASELINE.json configs and neighbouring shapes; prints a markdown table (profiles/rNN_shape_table.md)."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [("configs[1] classic loss+grad", []), ("configs[2] simplified loss+grad", ["--kind", "simplified"]),
        ("classic, ragged lengths", ["--ragged"]), ("classic, time-major float32", ["--time-major"]),
        ("classic, bfloat16 time-major", ["--dtype", "bf16", "--time-major"]),
        ("classic, U=64", ["--U", "64"]), ("classic, U=256", ["--U", "256"]), ("classic, U=300", ["--U", "300"]),
        ("classic, V=32", ["--V", "32"]), ("classic, V=512", ["--V", "512"]), ("classic, V=1024", ["--V", "1024"]),
        ("classic, V=2048", ["--V", "2048"]), ("classic, V=4096 (B=64)", ["--V", "4096", "--B", "64"]),
        ("classic, V=8192 (B=32)", ["--V", "8192", "--B", "32"]), ("classic, U=512", ["--U", "512"]),
        ("classic, B=64", ["--B", "64"]), ("classic, B=128", ["--B", "128"]), ("classic, B=1024", ["--B", "1024"]),
        ("classic, T=4000 (B=64)", ["--T", "4000", "--B", "64"]),
        ("configs[4] classic dense Hessian", ["--hessian", "--steps", "10", "--warmup", "2"]),
        ("simplified dense Hessian", ["--hessian", "--kind", "simplified", "--steps", "10", "--warmup", "2"])]
print("| workload | pipeline | ms per call | utterances/s | algorithmic GB/s | fraction of 8 TB/s |")
print("|:--|:--|--:|--:|--:|--:|")
for name, extra in RUNS:
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "100", "--warmup", "10"] + extra
    out = subprocess.run(args, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    r = d["roofline"]
    print(f"| {name} ({d['config']['workload']}) | {r['kernel'].split(' ')[0]} | {d['ms_per_step']:.3f} | {d['value']:.0f} | "
          f"{r['achieved']:.0f} | {r['frac']:.3f} |", flush=True)
