import csv, sys
names = ["L0 32->32", "L0 192->32", "L1 64->64", "L1 320->64", "L2 128->128", "L2 512->128", "L3 256->256", "L3 768->256", "L4 512->512"]
for path in sys.argv[1:]:
    rows = [r for r in csv.DictReader(open(path)) if 'conv3x3_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ds = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    wg = [int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) for r in rows]
    print(path.split('/')[-2], " | ".join("%s wg%d %.1f" % (names[i // 6], wg[i], min(ds[i:i + 6])) for i in range(0, len(ds), 6)))
