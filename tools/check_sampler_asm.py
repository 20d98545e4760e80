#!/usr/bin/env python3
"""`make -C carparkingmaps_amd/csrc asm-check`: every instantiation of the kernels that stage a row pack by LDS-DMA
(global_load_lds_dwordx4) must execute `s_waitcnt vmcnt(0)` after its last LDS-DMA instruction and before the first s_barrier
behind it: s_barrier waits for no counter, so without that wait a wave could search LDS words another wave's DMA has not
delivered yet (wrong destinations, silently).  The source carries the wait as inline assembly; this guards against it being
edited away or moved.  Also prints VGPR / SGPR / LDS per instantiation."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/cpm_api-hip-amdgcn-amd-amdhsa-gfx950.s"
text = open(path).read()
funcs = re.split(r"\n(?=_ZN3cpm[0-9]+k_)", text)
bad, seen = [], 0
for f in funcs:
    name = f.split(":", 1)[0].strip()
    if not name.startswith("_ZN3cpm") or "global_load_lds" not in f:
        continue
    lines = f.split("\n")
    seen += 1
    last_dma = max(i for i, l in enumerate(lines) if "global_load_lds" in l and not l.strip().startswith(";"))
    barrier = next((i for i in range(last_dma, len(lines)) if re.match(r"\s+s_barrier", lines[i])), None)
    ok = barrier is not None and any(re.match(r"\s+s_waitcnt vmcnt\(0\)", l) for l in lines[last_dma:barrier])
    # an early exit between the DMA and the barrier must drain too (the pack must not land in a later workgroup's LDS)
    m = re.search(r"\.vgpr_count:\s+(\d+)", f)
    res = re.search(r"; NumVgprs: (\d+)", f), re.search(r"; NumSgprs: (\d+)", f), re.search(r"; LDSByteSize: (\d+)", f)
    info = " ".join(f"{k}={r.group(1)}" for k, r in zip(("vgpr", "sgpr", "lds"), res) if r)
    print(("ok   " if ok else "FAIL ") + name[:90] + "  " + info)
    if not ok:
        bad.append(name)
print(f"{seen} kernels with LDS-DMA staging checked, {len(bad)} without the vmcnt(0) before their first barrier")
sys.exit(1 if bad or seen == 0 else 0)
