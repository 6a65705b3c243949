#!/bin/bash
# tools/spillcheck.sh [extra hipcc flags]: does ndt_align_kernel<true,false> touch scratch memory inside its point loops?
# (the loops are recognised by their nine ds_read_u16 slot probes and the v_permlane32_swap of the unit reduction behind them)
cd /root/repo
tools/isa.sh build_tmp/align_chk.s "$@" > /dev/null
python3 - <<'PY'
import re
L = open("/root/repo/build_tmp/align_chk.s").read().splitlines()
probe = [i for i, l in enumerate(L) if "ds_read_u16" in l]
swap = [i for i, l in enumerate(L) if "v_permlane32_swap" in l]
scr = [i for i, l in enumerate(L) if "scratch_" in l]
# point loops: from 120 lines ahead of a burst of probes to the last permlane32_swap of the reduction behind it
loops = []
i = 0
while i < len(probe):
    j = i
    while j + 1 < len(probe) and probe[j + 1] - probe[j] < 40: j += 1
    if j - i >= 8:
        end = max([s for s in swap if probe[j] < s < probe[j] + 900] or [probe[j] + 600])
        loops.append((probe[i] - 120, end))
    i = j + 1
inside = [s for s in scr if any(a <= s <= b for a, b in loops)]
print("kernel lines %d, scratch ops %d, point loops %s, scratch ops inside the point loops: %d" % (len(L), len(scr), loops, len(inside)))
for s in inside[:10]: print("   ", s, L[s].strip())
PY
