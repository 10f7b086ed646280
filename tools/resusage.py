# Summarises hipcc -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile) into one line per kernel.
import re, sys, subprocess
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip().split()[0] for b in blocks]
try:
    dn = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
except Exception:
    dn = names
for b, d in zip(blocks, dn):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    d = d.replace('void dcz::', '').split('(')[0]
    print('%-60s VGPR %3s AGPR %2s SGPR %3s scratch %4s occ %s spillS %3s spillV %3s LDS %6s' % (
        d[:60], g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
        g('SGPRs Spill'), g('VGPRs Spill'), g(r'LDS Size \[bytes/block\]')))
