"""Turn the two rocprofv3 --pmc passes of tools/gpu_pmc_bench*.sh (FETCH_SIZE, WRITE_SIZE) into
profiles/rNN_gemm_pmc_traffic[_MODE].txt and profiles/rNN_gemm_traffic[_MODE].json (bytes per GEMM launch, per kernel instance).
Usage: pmc_traffic.py FETCH_DB WRITE_DB OUT_TXT OUT_JSON [MODE [BENCH_ARGS]]"""
import collections, hashlib, json, os, re, sqlite3, sys

def per_kernel(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    # a dispatch reports one row per (counter, dimension instance): sum the instances of a dispatch first
    agg = collections.defaultdict(lambda: [0, 0.0])
    try:
        rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by 1,2,3").fetchall()
        for k, cn, _, v in rows:
            if "gemm_bf16_kernel" in k:
                a = agg[(re.sub(r"\(anonymous namespace\)::|void |\(.*", "", k), cn)]
                a[0] += 1; a[1] += v
    except sqlite3.OperationalError:
        for k, cn, v in rows:
            if "gemm_bf16_kernel" in k:
                a = agg[(re.sub(r"\(anonymous namespace\)::|void |\(.*", "", k), cn)]
                a[0] += 1; a[1] += v
    return agg

fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
MODE = sys.argv[5] if len(sys.argv) > 5 else "recompute"
BARGS = sys.argv[6] if len(sys.argv) > 6 else "--mode recompute --steps 3 --warmup 0 --prewarm 1 --no-cpu-baseline --no-profile --infer-steps 0 --no-extras --no-other-mode"
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (two separate passes) of",
         f"#   python bench.py {BARGS}   (MI355X; training mode {MODE})",
         "# per-dispatch averages for the bf16 GEMM kernels; counters are in KiB; gfx950 correction: FETCH_SIZE reports half",
         "# of a wide coalesced stream (MI355X_MICROARCH.md §HBM) -> HBM-side bytes = 2*FETCH_SIZE + WRITE_SIZE.",
         "# FETCH_SIZE counts fabric requests of the L2s (Infinity-Cache hits included), not only HBM reads.",
         f"# {'kernel':<62} {'counter':<12} {'launches':>8} {'avg_KiB':>12}"]
fw = {"F": [0, 0.0], "W": [0, 0.0]}
al = {"F": [0, 0.0], "W": [0, 0.0]}
for tag, agg in (("F", fetch), ("W", write)):
    for (k, cn), (n, tot) in sorted(agg.items()):
        lines.append(f"{k:<64} {cn:<12} {n:>8d} {tot / n:>12.1f}")
        al[tag][0] += n; al[tag][1] += tot                 # every bf16 GEMM launch (256-wide tile full or cut off, 128 x 128), every layout and epilogue
        if re.search(r"true, true, [025], [46], \d>", k):  # forward NT launches (store, residual and RoPE epilogues)
            fw[tag][0] += n; fw[tag][1] += tot
f_avg, w_avg = fw["F"][1] / max(fw["F"][0], 1), fw["W"][1] / max(fw["W"][0], 1)
fa_avg, wa_avg = al["F"][1] / max(al["F"][0], 1), al["W"][1] / max(al["W"][0], 1)
total, total_all = (2 * f_avg + w_avg) * 1024, (2 * fa_avg + wa_avg) * 1024
lines.append(f"# forward (NT) GEMM launches: avg FETCH_SIZE {f_avg:.0f} KiB, WRITE_SIZE {w_avg:.0f} KiB -> corrected traffic {total/1e9:.3f} GB per launch")
lines.append(f"# ALL bf16 GEMM launches (NT + NN + TN): avg FETCH_SIZE {fa_avg:.0f} KiB, WRITE_SIZE {wa_avg:.0f} KiB -> corrected traffic {total_all/1e9:.3f} GB per launch")
open(sys.argv[3], "w").write("\n".join(lines) + "\n")
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "navillm_amd", "csrc", "gemm_bf16.hip")
json.dump({"gemm_source_sha256": hashlib.sha256(open(_src, "rb").read()).hexdigest(),     # bench.py refuses the file when the kernel changed since
           "hbm_bytes_per_forward_gemm_launch": int(total), "hbm_bytes_per_gemm_launch_all_layouts": int(total_all),
           "fetch_kib_avg": f_avg, "write_kib_avg": w_avg, "launches_forward": fw["F"][0], "launches_all": al["F"][0],
           "mode": MODE, "bench_args": BARGS,
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, gfx950 FETCH_SIZE half-count correction", "source": sys.argv[3]}, open(sys.argv[4], "w"))
print(lines[-2]); print(lines[-1])
