import re,sys
txt=open(sys.argv[1]).read()
blocks=txt.split('  - .agpr_count:')
for b in blocks[1:]:
    name=re.search(r'\.name:\s+(\S+)',b).group(1)
    if 'trace_kernel' not in name: continue
    g=lambda k: re.search(r'\.%s:\s+(\d+)'%k,b).group(1)
    print(name[30:56], 'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'spill',g('vgpr_spill_count'),'scratch',g('private_segment_fixed_size'),'lds',g('group_segment_fixed_size'))
