"""profiles/pmc_traffic.json (the `roofline.traffic` figures bench.py reports) from a pmc_summary.txt written by
scripts/profile_round.sh: HBM / fabric bytes per launch = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE,
both in KiB.  Usage: python scripts/pmc_traffic.py profiles/r04/pmc_summary.txt > profiles/pmc_traffic.json"""
import json
import re
import sys

# round 4: QKV on the 192 x 384 tile stream (mode 2 = bias -> planes), FC1 on the 256 x 256 LDS-direct tiles, proj / FC2 on the stream's
# LayerNorm-fused mode 0 (averaged over its proj and FC2 launches)
KERNELS = {"attention": "attn_f16x3_pipe_kernel<true, true, false>", "gemm_qkv": "gemm_rowln16_kernel<RlGeo<192, 2>, 2, false>",
           "gemm_fc1_gelu": "gemm_plain256_kernel<1, true, 4, true, 0>", "gemm_proj": "gemm_rowln16_kernel<RlGeo<192, 2>, 0, false>"}
src = sys.argv[1]
blocks, cur = {}, None
for line in open(src):
    if not line.startswith(" "):
        cur = line.split("  (avg")[0].strip()
        blocks[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
        if m and cur:
            blocks[cur][m.group(1)] = float(m.group(2))
out = {"_comment": f"HBM/fabric bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes), "
                   f"scripts/profile_round.sh -> scripts/prof_forward.py 64 images, default precision f16x3; derived from {src} by scripts/pmc_traffic.py "
                   "(gemm_proj: the LN-fused kernel, averaged over its proj and FC2 launches)"}
for key, name in KERNELS.items():
    b = blocks.get(name)
    if b and "FETCH_SIZE" in b and "WRITE_SIZE" in b:
        out[key] = int(round((2 * b["FETCH_SIZE"] + b["WRITE_SIZE"]) * 1024))
print(json.dumps(out, indent=1))
