"""Vector-instruction budget of the match kernel's point loop from its ISA (tools/isa.sh writes build_tmp/align.s):
instructions per ROUND of 64 points by phase, and per 64 point-evaluations with the measured turns of the pair loop.
Usage: python tools/isa_budget.py [align.s] [turns_of_the_pair_loop_per_round = 3.7] [rounds_per_unit = 2.5]

The phases are found by landmarks of the first (solo-pass) copy of the loop: the prefetch load at the loop's head, the nine
ds_read_u16 slot probes, the nine ds_read_b64 centroid reads, the v_cmp_gt_f32 radius tests, the v_ffbl_b32 pair loop of the LDS
path, the v_cvt_f64_f32 of the untransformed point (finish_point) and the v_permlane32_swap butterfly."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build_tmp/align.s"
turns = float(sys.argv[2]) if len(sys.argv) > 2 else 3.7
rounds_per_unit = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5
L = open(path).read().splitlines()
is_valu = lambda l: re.match(r"\s+v_", l) is not None
is_lds = lambda l: re.match(r"\s+ds_", l) is not None
is_salu = lambda l: re.match(r"\s+s_(?!waitcnt|nop|cbranch|branch|barrier|sleep)", l) is not None


def count(a, b):
    seg = L[a:b]
    return sum(map(is_valu, seg)), sum(map(is_lds, seg)), sum(map(is_salu, seg))


probe = [i for i, l in enumerate(L) if "ds_read_u16" in l]
# first burst of nine probes
p0 = next(i for k, i in enumerate(probe) if k + 8 < len(probe) and probe[k + 8] - i < 30)
# loop head: the prefetch global load in front of the probes
head = max(i for i in range(p0 - 80, p0) if "global_load_dwordx2" in L[i])
cent_last = max(i for i in range(p0, p0 + 60) if "ds_read_b64" in L[i])
ffbl = [i for i in range(p0, p0 + 700) if "v_ffbl_b32" in L[i]]
cmp_first = next(i for i in range(cent_last, cent_last + 200) if "v_cmp_gt_f32" in L[i] or "v_cmp_ge_f32" in L[i])
# the LDS-path pair loop is the ffbl loop whose body reads ds_read2_b64 records
pair = next(i for i in ffbl if any("ds_read2_b64" in L[j] for j in range(i, i + 20)))
pair_end = next(i for i in range(pair, pair + 90) if "s_cbranch_execnz" in L[i])
radius_begin = max(i for i in range(cent_last, pair) if "s_cbranch_execz" in L[i] and i < cmp_first) if any("s_cbranch_execz" in L[i] for i in range(cent_last, cmp_first)) else cent_last
fin = next(i for i in range(pair_end, pair_end + 400) if "v_cvt_f64_f32" in L[i] and any("v_mul_f64" in L[j] for j in range(i + 1, i + 5)))
fin_end = next(i for i in range(fin, fin + 80) if "s_or_b64 exec" in L[i])
swap = next(i for i in range(fin_end, fin_end + 800) if "v_permlane32_swap" in L[i])
swap_end = next(i for i in range(swap, swap + 140) if "ds_write_b64" in L[i])
phases = [
    ("loop head: prefetch, transform, voxel index, window test, probe addresses", head, cent_last + 1),
    ("spill test + radius tests of the nine centroids, the pair loop's set-up", cent_last + 1, pair),
    ("pair loop, one turn", pair, pair_end + 1),
    ("finish_point (the point's ten sums)", fin, fin_end),
    ("unit butterfly (per unit, not per round)", swap - 1, swap_end + 1),
]
print("ISA %s; landmarks: head %d, probes %d, pair loop %d-%d, finish %d, butterfly %d" % (path, head, p0, pair, pair_end, fin, swap))
tot = 0.0
rows = []
for name, a, b in phases:
    v, l, s = count(a, b)
    mult = turns if name.startswith("pair") else (1.0 / rounds_per_unit if name.startswith("unit") else 1.0)
    rows.append((name, v, l, s, mult, v * mult))
    tot += v * mult
print("%-78s %6s %5s %5s %7s %9s %6s" % ("phase", "VALU", "LDS", "SALU", "x", "per round", "share"))
for name, v, l, s, mult, w in rows:
    print("%-78s %6d %5d %5d %7.2f %9.1f %5.1f%%" % (name, v, l, s, mult, w, 100 * w / tot))
print("%-78s %6s %5s %5s %7s %9.1f" % ("sum per round of 64 point-evaluations (pass loop only)", "", "", "", "", tot))
