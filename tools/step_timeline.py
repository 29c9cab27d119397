"""Per-queue timeline of ONE training step out of a rocprofv3 --kernel-trace CSV (steady state: the last full step).
   python tools/step_timeline.py gpurun_out/kt2/kt_kernel_trace.csv [step_index_from_end]
A step starts at a collate_fill_kernel dispatch.  Prints start (us, relative), duration, gap to the previous kernel on the
same queue, queue id and a short kernel name; then totals per queue (busy / gaps)."""
import csv, re, sys
from collections import defaultdict

def short(n):
    n = re.sub(r'^void ', '', n)
    n = n.replace('esc::', '').replace('dma::', '')
    m = re.match(r'([A-Za-z0-9_]+)(<[^(]*>)?', n)
    base = m.group(1) if m else n[:40]
    t = (m.group(2) or '') if m else ''
    t = t.replace('true', 'T').replace('false', 'F').replace(' ', '')
    return (base + t)[:70]

def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = [r for r in csv.DictReader(open(path)) if r['Kind'] == 'KERNEL_DISPATCH']
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    starts = [i for i, r in enumerate(rows) if 'collate_fill_kernel' in r['Kernel_Name']]
    if len(starts) < back + 1:
        print('not enough steps'); return
    a, b = starts[-back - 1], starts[-back]
    step = rows[a:b]
    t0 = int(step[0]['Start_Timestamp'])
    print(f"step of {len(step)} dispatches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us start to next start")
    last_end = {}
    busy = defaultdict(float); gaps = defaultdict(float); count = defaultdict(int)
    for r in step:
        q = r['Queue_Id']; s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        busy[q] += (e - s) / 1e3; gaps[q] += max(gap, 0.0); count[q] += 1
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f} q{q} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d}x{r['Workgroup_Size_X']:>4} lds {int(r['LDS_Block_Size']) // 1024:3d}K  {short(r['Kernel_Name'])}")
    for q in busy:
        print(f"queue {q}: {count[q]} kernels, busy {busy[q]:.1f} us, gaps {gaps[q]:.1f} us")

if __name__ == '__main__':
    main()
