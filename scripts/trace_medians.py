import csv,glob,statistics,sys
def med(tag):
    f=glob.glob(f'gpurun_out/{tag}/stats/**/*kernel_trace.csv',recursive=True)[0]
    g={}
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'anonymous' not in n or 'at::native' in n: continue
        n=n.replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        grid='x'.join(r[k] for k in ('Grid_Size_X','Grid_Size_Y','Grid_Size_Z'))
        g.setdefault((n,grid),[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    return g
for tag in sys.argv[1:]:
    print('==',tag)
    for k,v in sorted(med(tag).items()):
        if len(v) < 3 and 'forward_kernel' in k[0]: continue
        print(f'{k[0][:44]:46s} {k[1]:>16s} n={len(v):4d} med={statistics.median(v):9.1f} min={min(v):9.1f}')
