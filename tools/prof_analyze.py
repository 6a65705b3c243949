"""Diagnostic: per-scan summary of an NDT_PROF_DUMP file (see tools/prof_phases.py)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_dump_batch.bin", dtype=np.uint64)
PASSES = 24                                  # kProfPasses: shared-pass timelines behind the 32 words per scan
B = raw.size // (32 + PASSES * 16)
hp = raw[:8 * B].reshape(B, 8)
p2 = raw[8 * B:16 * B].reshape(B, 8).astype(np.float64)
stamps = raw[16 * B:32 * B].reshape(B, 16).astype(np.float64) * 0.01
t_eval = hp[:, 0] * 0.01; t_adv = (hp[:, 1] & 0xFFFFFFFF) * 0.01; t_fit = ((hp[:, 1] >> 32) & 0x7FFFFFFF) * 0.01
t0 = (hp[:, 2] & 0xFFFFFFFF) * 0.01
ev = hp[:, 3] & 0xFFFF; nsh = (hp[:, 3] >> 16) & 0xFFFF; nhu = (hp[:, 3] >> 32) & 0x7FFFFFFF
att = hp[:, 4] * 0.01; tw = hp[:, 6] * 0.01; tend = hp[:, 7] * 0.01
pc = [10, 25, 50, 75, 90, 99, 100]
print("setup us  mean %.0f" % t0.mean())
print("end us    mean %.0f pcts" % tend.mean(), np.percentile(tend, pc).round())
print("t_fit us  mean %.0f pcts" % t_fit.mean(), np.percentile(t_fit, pc).round())
print("passes    mean %.1f pcts" % ev.mean(), np.percentile(ev, pc))
if (att > 0).any():
    print("first helper attach pcts", np.percentile(att[att > 0], pc).round(), "scans helped", (att > 0).sum())
print("owner work per scan (setup+eval+adv) mean %.0f" % (t_eval + t_adv + t0).mean())
print("finish histogram /100us", np.histogram(tend, bins=np.arange(0, 2000, 100))[0])
for b in np.argsort(-tend)[:8]:
    print("  scan %3d end %.0f passes %d shared %d fit %.0f attach %.0f wait %.0f" % (b, tend[b], ev[b], nsh[b], t_fit[b], att[b], tw[b]))
n = p2[:, 0].sum()
if n > 0:
    print("shared derivative passes %d: prologue %.2f own units %.2f wait %.2f combine %.2f advance %.2f us (means)" %
          (n, p2[:, 1].sum() / n * 0.01, p2[:, 2].sum() / n * 0.01, p2[:, 3].sum() / n * 0.01, p2[:, 4].sum() / n * 0.01, p2[:, 5].sum() / n * 0.01))
raw2 = raw[8 * B:16 * B].reshape(B, 8)
print("setup phases us (means): region %.1f sort %.1f fill %.1f (marking %.1f, numbering done at %.1f)" % (
    (raw2[:, 6] >> 32).mean() * 0.01, (raw2[:, 6] & 0xFFFFFFFF).mean() * 0.01,
    (raw2[:, 7] >> 32).mean() * 0.01, ((raw2[:, 7] >> 16) & 0xFFFF).mean() * 0.01, (raw2[:, 7] & 0xFFFF).mean() * 0.01))
# round 5: the window's plan (7-9) runs inside the ordering, behind the offsets (3); 13 = record lines asked for; 4 = places handed
# out (the waves' turns) + entries written; 11 = order checked; 12 = (repaired) ; 5 = staged + copied out; 6 = published;
# 14 = record loads issued; 15 = slot table written; 10 = records in LDS
order = [(0, "scan loaded, optimiser start"), (1, "region"), (2, "histogram"), (3, "bitmap + offsets"),
         (4, "places (waves in turn) + entries"), (11, "order checked"), (5, "image of the ordered copy in LDS"),
         (7, "plan: occupancy"), (8, "plan: dilate"), (9, "plan: number"), (6, "copy-out issued, geometry stored (published)"),
         (14, "fetch: slot table + cells"), (15, "fetch: loads issued"), (10, "fetch: records in LDS")]
m = stamps.mean(0)
print("setup stamps us (mean, in program order: cumulative -> delta):")
prev = 0.0
for k, nm in order:
    if m[k] == 0:
        continue
    print("  %-34s %6.1f  (+%.1f)" % (nm, m[k], m[k] - prev)); prev = m[k]

# ---- timelines of the shared passes (absolute 100 MHz ticks): opened, own units done, collected, h, (seen, done) x 6
tl = raw[32 * B:].reshape(B, PASSES, 16).astype(np.int64)
ok = (tl[:, :, 0] > 0) & (tl[:, :, 2] > 0)
print("shared passes with a timeline:", int(ok.sum()))
rows = []
for b, k in zip(*np.nonzero(ok)):
    t = tl[b, k]; h = int(t[3])
    seen = [t[4 + 2 * r] - t[0] for r in range(min(h, 6)) if t[4 + 2 * r] > 0]
    done = [t[5 + 2 * r] - t[0] for r in range(min(h, 6)) if t[5 + 2 * r] > 0]
    if not seen or not done:
        continue
    rows.append((h, (t[1] - t[0]) * 0.01, (t[2] - t[0]) * 0.01, min(seen) * 0.01, max(seen) * 0.01, max(done) * 0.01,
                 np.mean([d - s for s, d in zip(seen, done)]) * 0.01))
rows = np.array(rows)
if len(rows):
    print("  h | passes | owner done | collected | helper sees (first/last) | last helper done | helper compute | collected - last done")
    for h in sorted(set(rows[:, 0].astype(int))):
        r = rows[rows[:, 0] == h]
        print("  %2d | %5d | %6.2f | %6.2f | %5.2f / %5.2f | %6.2f | %6.2f | %5.2f" % (
            h, len(r), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].mean(), r[:, 5].mean(), r[:, 6].mean(),
            (r[:, 2] - r[:, 5]).mean()))
