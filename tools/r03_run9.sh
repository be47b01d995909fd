set -e
cd /root/repo
for mi in 0 1 2; do echo "== MI=$mi"; FQSS_X3_MI=$mi python tools/kprobe.py time 2>&1 | grep "coded" | grep bwd_x; done > gpurun_out/r9_fwd.txt
python - <<'PY'
import re,collections
t=collections.defaultdict(dict); cur=None
for l in open('gpurun_out/r9_fwd.txt'):
    if l.startswith('=='): cur=l.strip()[3:]; continue
    m=re.match(r'(.*\S)\s+([\d.]+) us',l)
    t[m.group(1)][cur]=float(m.group(2))
cols=sorted(next(iter(t.values())).keys())
print(' '*32+' '.join(c.rjust(10) for c in cols))
for k,v in t.items(): print(k.ljust(32)+' '.join(f'{v[c]:10.1f}' for c in cols))
PY
