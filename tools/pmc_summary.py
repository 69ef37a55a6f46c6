import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
pat = sys.argv[2]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        d[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        d[r['Dispatch_Id']]['_dur_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        d[r['Dispatch_Id']]['_vgpr'] = r['VGPR_Count']; d[r['Dispatch_Id']]['_lds'] = r['LDS_Block_Size']
keys = sorted(d, key=int)
k = keys[len(keys) // 2]
print({kk: (int(v) if isinstance(v, float) else v) for kk, v in d[k].items()})
