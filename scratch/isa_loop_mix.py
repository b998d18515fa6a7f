"""Instruction mix of the hottest loop (the innermost backward-branch region with the most MFMAs) of one kernel in a hipcc
--save-temps assembly file.   usage: isa_loop_mix.py <file.s> <mangled kernel name>"""
import re, sys
src, kern = sys.argv[1], sys.argv[2]
lines, on, out = open(src).read().split("\n"), False, []
for l in lines:
    if l.startswith(kern + ":"):
        on = True
    if on:
        out.append(l)
        if "s_endpgm" in l:
            break
labels = {m.group(1): i for i, l in enumerate(out) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
best = None
for i, l in enumerate(out):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        body = out[labels[m.group(1)]:i + 1]
        n = sum("v_mfma" in x for x in body)
        if n and (best is None or n / len(body) > best[0] / len(best[1])):
            best = (n, body)
n, body = best
c = lambda p: sum(1 for x in body if re.search(p, x))
valu = sum(1 for x in body if re.match(r"^\s+v_", x) and not re.search("v_mfma|v_exp_f32|v_cvt_pk_bf16", x))
salu = sum(1 for x in body if re.match(r"^\s+s_", x) and not re.search("s_waitcnt|s_barrier|s_nop", x))
print(f"{kern}: hot loop {len(body)} lines: MFMA {n}, v_exp_f32 {c('v_exp_f32')}, v_cvt_pk_bf16 {c('v_cvt_pk_bf16')}, other VALU {valu}, "
      f"ds_read {c('ds_read')} (of them transposing {c('ds_read_b64_tr')}), s_waitcnt {c('s_waitcnt')}, s_barrier {c('s_barrier')}, SALU {salu}")
