import sys,re
f=sys.argv[1]; show=len(sys.argv)>2
s=open(f).read().split('\n')
start=[i for i,l in enumerate(s) if l.startswith('_ZN2rt16render_kernel_wsILb1ELi512ELb0E')][0]
end=[i for i,l in enumerate(s) if i>start and l.startswith('.Lfunc_end')][0]
# lines that belong to depth-3 loop: blocks annotated "in Loop: Header=BBx Depth=3" or the header
inloop=False; body=[]
for i in range(start,end):
    l=s[i]
    if re.match(r'^(\.LBB|; %bb)',l):
        inloop = ('Depth=3' in l) or (i+3<end and any('Inner Loop Header: Depth=3' in s[k] for k in range(i,i+4)))
    if inloop: body.append(l)
ins=[l for l in body if l.startswith('\t') and not l.strip().startswith(';')]
print('instrs',len(ins),'v_mov',sum('v_mov' in l for l in ins),'valu',sum(l.strip().startswith('v_') for l in ins),'salu',sum(l.strip().startswith('s_') for l in ins),'ds',sum(l.strip().startswith('ds_') for l in ins))
for l in s[start:end+200]:
    if re.search(r'\.vgpr_count|; NumVgprs|; Occupancy|ScratchSize',l): print(l.strip())
if show: print('\n'.join(body))
