"""One frame of a rocprofv3 --kernel-trace run, kernel by kernel: queue, grid, start (us from the frame's first kernel) and duration —
shows which launches overlap (round 6: the row-range chains of the network).
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python tools/frame_time.py 66048 3 1; python tools/frame_trace.py /tmp/tr"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: find last backend kernel and the 40 kernels before it
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "pn_backend_kernel" in n]
end = idx[-1]; start = idx[-2] + 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:end + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    print(f"{n:40s} q={r.get('Queue_Id','?'):>3s} grid={r.get('Grid_Size', r.get('Grid_Size_X','?')):>8s} start={(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
