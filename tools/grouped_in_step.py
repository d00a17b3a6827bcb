import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
adam=[i for i,r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name'] and 'FusedOptimizer' in r['Kernel_Name']]
a,b=adam[-2],adam[-1]
win=rows[a+1:b+1]
t0=int(win[0]['Start_Timestamp'])
for r in win:
    n=r['Kernel_Name']
    if 'grouped_gemm' in n:
        print("grouped at %8.1f us  dur %6.1f us  grid %s" % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X','?')+"x"+r.get('Grid_Size_Y','?')))
