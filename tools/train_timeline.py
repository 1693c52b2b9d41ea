"""Timeline of ONE training iteration from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o train -- python3 tools/bench_train.py --steps 3 --warmup 1
    python tools/train_timeline.py OUT/.../train_kernel_trace.csv
Per millisecond: busy time of the main stream by kernel family, busy time of the side streams; then the main stream's
gaps (> 15 us) with the kernels on either side -- host syncs and stream joins show up here, not in the kernel statistics."""
import collections, csv, sys

FAMILIES = ['conv_igemm_big', 'wgrad_fold', 'stem_wgrad', 'wgrad_kernel', 'bn_reduce', 'bn_bwd_apply', 'bn_bwd_finalize',
            'bn_apply', 'bn_fwd_finalize', 'bn_dual', 'bn_act_mask', 'pack_weight', 'limb_dual', 'limb_kernel', 'unary',
            'head_grad', 'head_bias', 'conv_igemm_kernel', 'stem', 'conv64', 'adam', 'relu_mask', 'add_relu', 'elementwise',
            'reduce_kernel', 'Fill', 'sumsq', 'colsum', 'copyBuffer', 'Cat', 'gradnorm']


def short(n):
    for k in FAMILIES:
        if k in n:
            return k
    return n[:25]


rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
it = rows[ad[-2] + 1:ad[-1] + 1]
T0, T1 = it[0]['s'], it[-1]['e']
qs = collections.Counter(r['Queue_Id'] for r in it)
mq = max(qs, key=lambda q: qs[q])
main = [r for r in it if r['Queue_Id'] == mq]
print(f"iteration {(T1 - T0) / 1e6:.3f} ms, {len(it)} kernels, per queue {dict(qs)}")
for q in qs:
    sub = [r for r in it if r['Queue_Id'] == q]
    print(f"  queue {q}: busy {sum(r['e'] - r['s'] for r in sub) / 1e6:.3f} ms, first {(sub[0]['s'] - T0) / 1e6:.2f}, last {(sub[-1]['e'] - T0) / 1e6:.2f}")
for b in range(int((T1 - T0) / 1e6) + 1):
    lo = T0 + b * 1e6; hi = lo + 1e6
    cat = collections.Counter()
    for r in main:
        o = min(r['e'], hi) - max(r['s'], lo)
        if o > 0:
            cat[short(r['Kernel_Name'])] += o / 1e3
    side = sum(max(0, min(r['e'], hi) - max(r['s'], lo)) for r in it if r['Queue_Id'] != mq) / 1e3
    print(f"{b:2d} ms  main {sum(cat.values()):5.0f} us  side {side:5.0f} us | " + ' '.join(f"{k}:{v:.0f}" for k, v in cat.most_common(6)))
gaps = [(round((a['e'] - T0) / 1e6, 2), round((b['s'] - a['e']) / 1e3), short(a['Kernel_Name']), short(b['Kernel_Name']))
        for a, b in zip(main, main[1:]) if b['s'] - a['e'] > 15000]
print(f"main-stream gaps: {sum(max(0, b['s'] - a['e']) for a, b in zip(main, main[1:])) / 1e6:.3f} ms in all; {len(gaps)} above 15 us = {sum(g[1] for g in gaps)} us:")
for g in gaps:
    print("   at %.2f ms: %4d us  after %s, before %s" % g)

if len(sys.argv) > 2 and sys.argv[2] == '--small':
    # every launch of the named families on the main stream with its neighbours: where do the copies / repacks / tiny kernels sit?
    fam = sys.argv[3].split(',') if len(sys.argv) > 3 else ['copyBuffer', 'pack_weight', 'elementwise', 'reduce_kernel', 'Fill', 'Cat']
    per = collections.Counter(); tot = collections.Counter()
    for r in it:
        k = short(r['Kernel_Name']); per[(k, r['Queue_Id'] == mq)] += 1; tot[(k, r['Queue_Id'] == mq)] += (r['e'] - r['s']) / 1e3
    print("launches per iteration by family (main stream / other streams):")
    for (k, m), n in sorted(per.items(), key=lambda kv: -tot[kv[0]]):
        print(f"   {k:26s} {'main ' if m else 'other'} {n:4d} launches {tot[(k, m)]:8.1f} us")
    print("named families, in order, with the launch before and after on the same queue:")
    byq = collections.defaultdict(list)
    for r in it:
        byq[r['Queue_Id']].append(r)
    for q, sub in byq.items():
        for i, r in enumerate(sub):
            k = short(r['Kernel_Name'])
            if k in fam:
                a = short(sub[i - 1]['Kernel_Name']) if i else '-'
                b = short(sub[i + 1]['Kernel_Name']) if i + 1 < len(sub) else '-'
                gap = (r['s'] - sub[i - 1]['e']) / 1e3 if i else 0
                print(f"   q{q} {(r['s'] - T0) / 1e6:7.3f} ms {k:14s} {(r['e'] - r['s']) / 1e3:6.1f} us (gap before {gap:5.1f}) after {a}, before {b}")
