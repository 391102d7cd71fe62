#!/usr/bin/env python3
"""Opcode histogram of gfx950 device functions from the disassembly of a built library (no GPU needed).

usage: tools/valu_histogram.py [lib.so] [--kernels 'regex'] [--trip N] [--loops]

For every selected device function: instructions by class, (a) static, (b) weighted by loop nesting with an assumed trip count
per level (--trip, default 8: the hot loops of these kernels run 4-25 iterations), and the same split for the VALU classes
only - the decomposition of what the SQ counters call "other" (moves, selects, bit operations, AGPR copies, lane exchanges,
compares) next to integer address arithmetic and real floating-point math.  --loops lists every natural loop (backward
branch) with its own counts so that a known trip count can be applied by hand.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT = (r"mbf_kernel<true, 5, 1, 16, false, 0>|xbf_kernel<true, 3, 2, 8, 8, [123], (32|24|288), \d+>|sep_kernel<true, 0, false, true>|"
           r"chain_kernel<true, 0>|pw_gemm_kernel<1, 2, 2, 2, 0, 3, 4, true>")


def classify(op):
    o = op
    if o.startswith("v_mfma") or o.startswith("v_smfmac"):
        return "mfma"
    if o.startswith("v_"):
        if o.startswith(("v_accvgpr", "v_mov_b", "v_swap")):
            return "valu.move"
        if "dpp" in o or o.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")):
            return "valu.lane"
        if o.startswith("v_cndmask"):
            return "valu.select"
        if o.startswith("v_cmp"):
            return "valu.compare"
        if o.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
            return "valu.transcendental"
        if o.startswith("v_cvt"):
            return "valu.convert"
        if o.startswith(("v_perm_b32", "v_lshlrev", "v_lshrrev", "v_ashrrev", "v_and_b", "v_or_b", "v_xor", "v_bfe", "v_bfi", "v_lshl_or", "v_and_or", "v_or3", "v_not",
                         "v_alignb", "v_lshl_b", "v_lshr_b", "v_ashr", "v_pk_lshl", "v_pk_lshr", "v_bfm", "v_ffb", "v_bcnt", "v_mbcnt")):
            return "valu.bitop"
        if re.match(r"v_(add|sub|subrev|mul_lo|mul_hi|mul_u|mul_i|mad_u|mad_i|add3|lshl_add|add_lshl|addc|subb|min_[iu]|max_[iu]|med3_[iu]|pk_add_[iu]|pk_mul_lo|pk_mad_[iu]|pk_sub_[iu])", o) and not re.search(r"_f(16|32|64)|bf16", o):
            return "valu.int"
        if o.startswith("v_pk_"):
            return "valu.fp_packed"
        if o.startswith(("v_dot", "v_fma", "v_fmac", "v_mul_f", "v_add_f", "v_sub_f", "v_max_f", "v_min_f", "v_mad_f", "v_med3_f", "v_ldexp", "v_frexp", "v_div", "v_rndne", "v_floor",
                         "v_fract", "v_trunc", "v_ceil", "v_mac_f", "v_max3", "v_min3", "v_mul_legacy", "v_subrev_f", "v_fmaak", "v_fmamk")):
            return "valu.fp"
        if o.startswith("v_nop"):
            return "valu.nop"
        return "valu.other"
    if o.startswith("ds_"):
        return "lds"
    if o.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if o.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem"
    if o.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_sethalt")):
        return "wait"
    if o.startswith("s_barrier"):
        return "barrier"
    if o.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if o.startswith("s_"):
        return "salu"
    return "?"


def code_objects(lib):
    d = tempfile.mkdtemp(prefix="hepdis")
    tmp = os.path.join(d, os.path.basename(lib))
    subprocess.run(["cp", lib, tmp], check=True)           # (the extractor writes next to its input)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if "gfx950" in f]


def functions(co):
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    cur, out = None, {}
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = m.group(2); out[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def analyse(insts, trip):
    addrs = [a for a, _, _ in insts]
    loops = []                                           # (head address, tail address)
    for i, (a, op, args) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"(-?\d+)\s*$", args.split()[-1]) if args else None
            # llvm-objdump prints the target as a symbol+offset comment-less operand; recompute from the simm16
            try:
                simm = int(args.split()[-1])
            except ValueError:
                continue
            if simm >= 32768:
                simm -= 65536
            target = a + 4 + 4 * simm
            if target <= a:
                loops.append((target, a))
    # one loop per head (several backward branches to the same head = `continue`s of one loop); nesting capped at 3 levels
    by_head = {}
    for h, t in loops:
        by_head[h] = max(by_head.get(h, t), t)
    loops = sorted(by_head.items())
    depth = []
    for a in addrs:
        depth.append(min(3, sum(1 for h, t in loops if h <= a <= t)))
    return loops, depth


def main():
    argv = sys.argv[1:]
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hmd_ego_pose_amd", "libhep.so")
    pat, trip, show_loops, table, names_file = DEFAULT, 8, False, False, None
    while argv:
        a = argv.pop(0)
        if a == "--kernels": pat = argv.pop(0)
        elif a == "--trip": trip = int(argv.pop(0))
        elif a == "--loops": show_loops = True
        elif a == "--table": table = True
        elif a == "--names": names_file = argv.pop(0)
        else: lib = a
    rx = re.compile(pat)
    exact = set(l.strip() for l in open(names_file)) if names_file else None
    rows = []
    for co in code_objects(lib):
        fns = functions(co)
        names = subprocess.run(["c++filt"], input="\n".join(fns.keys()), capture_output=True, text=True).stdout.splitlines()
        for mangled, name in zip(fns.keys(), names):
            name = re.sub(r"^void ", "", name)
            if exact is not None:
                if re.sub(r"\(.*\)$", "", name) not in exact:
                    continue
            elif not rx.search(name):
                continue
            insts = fns[mangled]
            if table:      # one line per device function: static instruction counts by unit (before / after comparisons of a build)
                cs = collections.Counter(classify(op).split(".")[0] for _, op, _ in insts)
                rows.append((re.sub(r"\(.*\)$", "", name), len(insts), cs["valu"], cs["salu"], cs["vmem"], cs["lds"], cs["mfma"]))
                continue
            loops, depth = analyse(insts, trip)
            stat, wgt = collections.Counter(), collections.Counter()
            ops_w = collections.Counter()
            for (a, op, args), d in zip(insts, depth):
                c = classify(op)
                stat[c] += 1; wgt[c] += trip ** d
                if c.startswith("valu"):
                    ops_w[(c, re.sub(r"_(e32|e64|sdwa|dpp)$", "", op))] += trip ** d
            print(f"== {name}\n   {len(insts)} instructions, {len(loops)} loops (max nesting {max(depth) if depth else 0}); weighted = {trip}^depth")
            tv_s = sum(v for k, v in stat.items() if k.startswith("valu")); tv_w = sum(v for k, v in wgt.items() if k.startswith("valu"))
            print(f"   {'class':22s} {'static':>8s} {'%valu':>6s} {'weighted':>10s} {'%valu':>6s}")
            for k in sorted(stat, key=lambda k: -wgt[k]):
                isv = k.startswith("valu")
                print(f"   {k:22s} {stat[k]:8d} {100.0 * stat[k] / tv_s if isv else 0:6.1f} {wgt[k]:10d} {100.0 * wgt[k] / tv_w if isv else 0:6.1f}")
            math = sum(wgt[k] for k in ("valu.fp", "valu.fp_packed", "valu.transcendental"))
            print(f"   VALU total {tv_s} static / {tv_w} weighted; floating-point math + transcendentals = {100.0 * math / max(tv_w, 1):.1f} % of the weighted VALU stream")
            print("   top weighted VALU opcodes: " + ", ".join(f"{op} {100.0 * w / max(tv_w, 1):.1f}%" for (c, op), w in ops_w.most_common(14)))
            if show_loops:
                for h, t in sorted(loops):
                    cs = collections.Counter(classify(op) for a, op, _ in insts if h <= a <= t)
                    dd = sum(1 for h2, t2 in loops if h2 <= h and t <= t2)
                    print(f"   loop {h:#x}..{t:#x} depth {dd}: " + ", ".join(f"{k} {v}" for k, v in cs.most_common()))
            print()
    if table:
        _print_table(rows)


def _print_table(rows):
    print(f"{'device function':52s} {'instr':>6s} {'valu':>6s} {'salu':>6s} {'vmem':>5s} {'lds':>5s} {'mfma':>5s}")
    for r in sorted(rows):
        print(f"{r[0]:52s} {r[1]:6d} {r[2]:6d} {r[3]:6d} {r[4]:5d} {r[5]:5d} {r[6]:5d}")
    print(f"{'total':52s} {sum(r[1] for r in rows):6d} {sum(r[2] for r in rows):6d} {sum(r[3] for r in rows):6d} {sum(r[4] for r in rows):5d} {sum(r[5] for r in rows):5d} {sum(r[6] for r in rows):5d}")


if __name__ == "__main__":
    main()
