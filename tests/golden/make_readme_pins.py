#!/usr/bin/env python3
"""The two printouts of the reference's README.md (a notebook export) that can pin anything on this path, transcribed into
fixtures (data only: the printed OUTPUT cells, not the notebook's code):

  readme_initializestates.json  README.md:505-530 -- `state_matrix` after the initial placement (src/initializestates.jl:11-17):
                                its shape, the rows Julia printed at the top and at the bottom, all other columns zero.
  readme_rawdata_rows.csv       README.md:259 -- the first 30 rows of the Melbourne Uber Movement file as the DataFrame printout
                                shows them (6 of the 7 columns; the seventh, geometric_standard_deviation_travel_time, was
                                "omitted printing" and is written as 1.0 here -- the path never reads it, src/createdatamatrix.jl:5).
  readme_rawdata_rows.json      the same rows as numbers + the notebook's stated sizes (2,357 zones, 11,566,494 rows, 91 % sparsity)

They pin inputs of the path only (a1 initial placement, f2 CSV reader / createdatamatrix, incl. the hod 0 -> 24 remap of row 11);
the sampler itself stays "parity unpinned" (the reference is unseeded and holds no outputs of it).

    python tests/golden/make_readme_pins.py [/root/reference/README.md]
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
readme = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/README.md"
lines = open(readme, encoding="utf-8").read().split("\n")

# ---- state_matrix printout
start = next(i for i, l in enumerate(lines) if re.match(r"\s*(\d+)×(\d+) Array\{Int64,2\}:", l))
m = re.match(r"\s*(\d+)×(\d+) Array\{Int64,2\}:", lines[start])
shape = [int(m.group(1)), int(m.group(2))]
top, bottom, seen_gap = [], [], False
for l in lines[start + 1:]:
    if "⋮" in l:
        seen_gap = True
        continue
    nums = l.split()
    if len(nums) != shape[1] or not all(re.fullmatch(r"\d+", x) for x in nums):
        if seen_gap and bottom:
            break
        continue
    (bottom if seen_gap else top).append([int(x) for x in nums])
cpz_line = next(l for l in lines if re.match(r"cars_per_zone = (\d+);", l))
cpz = int(re.match(r"cars_per_zone = (\d+);", cpz_line).group(1))
json.dump({"source": f"README.md:{start + 1}-{start + 1 + len(top) + len(bottom) + 1} (notebook output cell)", "shape": shape, "cars_per_zone": cpz,
           "first_rows": top, "last_rows": bottom}, open(os.path.join(HERE, "readme_initializestates.json"), "w"), indent=1)
print("state_matrix", shape, len(top), "top rows,", len(bottom), "bottom rows, cars_per_zone", cpz)

# ---- rawdata printout (an HTML table on one line)
tline = next(i for i, l in enumerate(lines) if "<table class=\"data-frame\">" in l and "sourceid" in l)
html = lines[tline]
header = re.findall(r"<th>([a-z_]+)</th>", html)
nrows_total, ncols_total = [int(x.replace(",", "")) for x in re.search(r"<p>([\d,]+) rows × (\d+) columns", html).groups()]
rows = []
for r in re.findall(r"<tr><th>(\d+)</th>((?:<td>[^<]*</td>)+)</tr>", html):
    rows.append([float(x) for x in re.findall(r"<td>([^<]*)</td>", r[1])])
assert header[:6] == ["sourceid", "dstid", "hod", "mean_travel_time", "standard_deviation_travel_time", "geometric_mean_travel_time"], header
assert [int(r[0]) for r in rows] and len(rows) == 30
full_header = header[:6] + ["geometric_standard_deviation_travel_time"]
with open(os.path.join(HERE, "readme_rawdata_rows.csv"), "w") as f:
    f.write(",".join(full_header) + "\n")
    for r in rows:
        f.write(f"{int(r[0])},{int(r[1])},{int(r[2])},{r[3]!r},{r[4]!r},{r[5]!r},1.0\n")
zones = int(next(re.search(r"N = ([\d,]+)", l).group(1).replace(",", "") for l in lines if "first N = " in l))
sp = int(next(re.search(r"Sp = (\d+)%", l).group(1) for l in lines if re.match(r"\s+The sparsity of the imported data set", l)))
json.dump({"source": f"README.md:{tline + 1} (DataFrame printout), sizes README.md:264,310", "columns_printed": header[:6], "rows": rows, "rows_total": nrows_total,
           "columns_total": ncols_total, "number_zones": zones, "sparsity_percent": sp}, open(os.path.join(HERE, "readme_rawdata_rows.json"), "w"), indent=1)
print("rawdata", len(rows), "rows of", nrows_total, "x", ncols_total, "; zones", zones, "; sparsity", sp, "%")
