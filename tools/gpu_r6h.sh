#!/bin/bash
# Round 6: kernel timeline of the trait-level harness: do gather and scatter overlap on the link?
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=$PWD/symphonia_amd/build/decoders_bench
cd /tmp
SYMACCEL_BATCH_COPY_WGS=256 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r6h -o dec -- $B --codec aac --streams 256 --lookahead 256 --packets 1024 --threads 16 --direct > $OUT/r06h_line.json 2> $OUT/r06h_err.txt
ls -R $OUT/prof_r6h | head
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/prof_r6h/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), 'kernel records; columns', list(rows[0].keys()))
ev = []
for r in rows:
    n = r['Kernel_Name']; s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    kind = 'copy' if 'batch_copy' in n else ('synth' if 'aac_synth' in n else ('flag' if 'flag' in n else 'other'))
    ev.append((s, e, kind, int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0), r.get('Queue_Id', ''), r.get('Stream_Id','')))
ev.sort()
t0 = ev[len(ev)//2][0]
# the last 60 % of the run: busy time per kind, overlap between copies
lo = ev[int(len(ev)*0.4)][0]; hi = ev[-1][1]
def busy(kind_pred):
    iv = sorted((max(s,lo), min(e,hi)) for s,e,k,*_ in ev if kind_pred(k) and e > lo)
    tot = 0; cur_s = cur_e = None
    for s,e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
span = hi - lo
print('span ms', span/1e6, 'copy busy', busy(lambda k: k=='copy')/span, 'synth busy', busy(lambda k: k=='synth')/span)
# concurrency of copy kernels: time with >= 2 copy kernels running
pts = []
for s,e,k,*_ in ev:
    if k == 'copy' and e > lo: pts += [(max(s,lo), 1), (min(e,hi), -1)]
pts.sort()
lvl = 0; last = lo; hist = collections.Counter()
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
print('copy-kernel concurrency (fraction of span):', {k: round(v/span,3) for k,v in sorted(hist.items())})
queues = collections.Counter((k, q) for s,e,k,g,q,st in ev)
print('kernels by (kind, queue):', dict(queues))
durs = collections.defaultdict(list)
for s,e,k,g,q,st in ev:
    if e > lo: durs[(k, g)].append((e-s)/1e3)
for key, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(key, 'n', len(v), 'avg us', round(sum(v)/len(v),1), 'total ms', round(sum(v)/1e3,2))
PY
cat $OUT/r06h_line.json | head -c 600
