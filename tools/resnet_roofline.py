"""Per-kernel roofline rows of the IMPALA-ResNet learner from a rocprofv3 kernel-stats table (tools/rocprof_summary.py over tools/rn_microbench.py):
executed flops of a 3840-frame minibatch per launch from the kernel's own geometry (template arguments CI, CO, H in its name), TFLOP/s and fraction
of the fp32 MFMA peak (157.3).  Learner-size launches only: a kernel that also runs at 120 frames inside the rollout (its shortest launch is far
below its average) is priced at its longest launch, one that only runs at 3840 frames at its average (the first, cold launch of each is 2-3x the rest
and would otherwise be reported as the kernel's time).
usage: python tools/resnet_roofline.py profiles/r04_resnet_kernel_stats.md"""
import re
import sys

PEAK, MB = 157.3, 3840
rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"\| (?:void )?(\w+)<?(.*?)>? \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", ln)
    if not m:
        continue
    name, targs, calls, total, avg, mn, mx = m.group(1), m.group(2), int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)), float(m.group(7))
    g = re.search(r"Rn\w*Geom<(\d+), (\d+), (\d+)", targs)
    if name in ("rn_rw_kernel", "rn_wgrad_kernel", "rn_wgrad2_kernel") and g:
        ci, co, h = (int(x) for x in g.groups())
        flops = 2.0 * MB * h * h * 9 * ci * co
        kind = "wgrad" if "wgrad" in name else ("dgrad" if re.search(r", (2|3|4), (true|false)$", targs) else "fwd")
    elif name in ("rn_conv0_pool_kernel", "rn_conv0_pool_reg_kernel"):   # strips (small batches) / one wave per frame, pooled on the accumulators
        ci, co, h, kind = 4, 16, 84, "fwd+pool"
        flops = 2.0 * MB * 84 * 84 * 9 * 4 * 16
    elif name == "rn_wgrad0_sparse_kernel":
        ci, co, h, kind = 4, 16, 84, "wgrad (sparse: pooled elements only)"
        flops = 2.0 * MB * 42 * 42 * 9 * 4 * 16      # executed: one of the four conv outputs under each pooled element carries gradient
    else:
        continue
    learner_us = mx if mn < 0.5 * avg else avg     # mixed 120-frame / 3840-frame launches: the learner-size ones are the long ones
    if learner_us < 60:
        continue
    rows.append((total, name, f"{ci}->{co} @ {h}x{h} {kind}", learner_us, flops))
rows.sort(reverse=True)
print("| kernel | layer | us per 3840-frame launch | executed GFLOP | TFLOP/s | of 157.3 |")
print("|---|---|---|---|---|---|")
for total, name, what, us, flops in rows[:16]:
    tf = flops / us / 1e6
    print(f"| {name} | {what} | {us:.1f} | {flops / 1e9:.1f} | {tf:.1f} | {tf / PEAK:.2f} |")
