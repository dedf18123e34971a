"""Per-kernel statistics of a hipcc --save-temps assembly file: registers, scratch, instruction counts of interest.
Usage: isa_stats.py FILE.s [substring filter ...]"""
import re, sys
s = open(sys.argv[1]).read()
filt = sys.argv[2:]
funcs = re.split(r"\n(?=\t\.section\t\.text\.)", s)
for f in funcs:
    m = re.search(r"^(_Z\S+):\s", f, re.M)
    if not m:
        continue
    name = m.group(1)
    if filt and not any(x in name for x in filt):
        continue
    def g(key):
        r = re.search(r"\.set " + re.escape(name) + r"\." + key + r", (\d+)", s)
        return r.group(1) if r else "?"
    body = f
    cnt = {k: len(re.findall(r"^\s+" + k + r"\b", body, re.M)) for k in
           ("v_mfma\w*", "v_cvt_pk_f32_fp8", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_cvt_scalef32_pk_bf16_fp8", "ds_read_b64", "ds_read_b128",
            "ds_read_b64_tr_b16", "buffer_load_dwordx4", "s_barrier", "scratch_\w+", "s_waitcnt vmcnt\(0\)")}
    short = re.sub(r"_ZN12_GLOBAL__N_1", "", name)
    print(short[:110], "vgpr", g("num_vgpr"), "agpr", g("num_agpr"), "scratch", g("private_seg_size"), {k: v for k, v in cnt.items() if v})
