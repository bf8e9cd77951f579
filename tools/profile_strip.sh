#!/bin/bash
# tools/profile_strip.sh TAG — kernel trace of ONE rank's share of the strong-scaling cut of config C4 (tools/bench_strip.py) on the
# GPU box: the sweep kernel on a 177-row block + 2 x depth redundant rows, with and without the state exchange (rank = its own
# neighbour).  Summary -> gpurun_out/profiles_TAG/TAG_strip.md (copy into profiles/).
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_strip_$TAG
mkdir -p $O $R/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/bench_strip.py --depth 8 16 32 > $O/strip.json 2> $O/strip.err || { tail -5 $O/strip.err; exit 1; }
python3 $R/tools/bench_strip.py --depth 8 16 32 --exchange > $O/strip_x.json 2> $O/strip_x.err || { tail -5 $O/strip_x.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_strip.py --depth 8 --sweeps 800 --exchange > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, json, collections
O="$O"; tag="$TAG"
dur=collections.defaultdict(list)
for f in glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows: dur[r["Kernel_Name"].split("(")[0].replace("void ","")[:70]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
lines=["# One rank of the strong-scaling cut of C4 on one MI355X ("+tag+")","",
       "\`tools/bench_strip.py\`: the 1415 x 1415 grid cut into 8 row blocks; this is block 3 (177 owned rows) with \`depth\` redundant rows per side.",
       "Sweeps go to the library in batches of \`depth\` (one cx_sweep call); with the exchange, each batch is preceded by pack, grouped RCCL send/recv (the rank is its own neighbour: a middle rank's volume) and unpack, all on the handle's stream.","",
       "| depth | exchange | us per sweep (wall) | us per sweep (device events) | whole grid us per sweep | ideal = whole / 8 | ratio to ideal | 8-GPU speed-up if every rank ran like this |","|---|---|---|---|---|---|---|---|"]
for fn in ("strip.json","strip_x.json"):
    for l in open(O+"/"+fn):
        l=l.strip()
        if not l.startswith("{"): continue
        d=json.loads(l)
        lines.append(f"| {d['depth']} | {'yes' if d['exchange'] else 'no'} | {d['us_per_sweep_wall']:.2f} | {d['us_per_sweep_device']:.2f} | {d['whole_grid_us_per_sweep']:.2f} | {d['ideal_us']:.2f} | {d['ratio_to_ideal']:.3f} | {d['speedup_if_all_ranks_like_this']:.2f} |")
lines+=["","## kernel trace (rocprofv3 --kernel-trace --stats; depth 8 with the exchange, 800 sweeps per timed repetition)","","| kernel | calls | avg us | median us |","|---|---|---|---|"]
for k,v in sorted(dur.items(), key=lambda kv:-sum(e-s for s,e in kv[1]))[:8]:
    d=sorted((e-s)/1e3 for s,e in v)
    lines.append(f"| \`{k}\` | {len(d)} | {sum(d)/len(d):.2f} | {d[len(d)//2]:.2f} |")
# gap between consecutive sweep launches on the strip
sw=[x for k,v in dur.items() if "k_sweep<" in k for x in v]
sw.sort()
gaps=sorted((sw[i+1][0]-sw[i][1])/1e3 for i in range(len(sw)-1) if 0 <= sw[i+1][0]-sw[i][1] < 20000)
if gaps: lines+=["",f"gap between the end of one sweep kernel and the start of the next (same stream, back to back): median {gaps[len(gaps)//2]:.2f} us over {len(gaps)} launches"]
open("gpurun_out/profiles_"+tag+"/"+tag+"_strip.md","w").write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
