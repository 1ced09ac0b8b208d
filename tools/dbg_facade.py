import sys, os, subprocess
ROOT='/root/repo'
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
import numpy as np, klang_amd
from scenario_io import Scenario
s=Scenario.load(ROOT+'/tests/golden/supersaw_poly.scn')
env=dict(os.environ, KLG_FACADE_DEBUG='1')
out=subprocess.run([ROOT+'/oracle/_ref/facade_supersaw', ROOT+'/tests/golden/supersaw_poly.scn','/tmp/f.bin'],env=env,capture_output=True,text=True).stdout.splitlines()
print('\n'.join(out[:4]))
A=klang_amd.SynthBank('supersaw',1,32,max_block=256)
n=0
for (b,t,sy,a,bb,seed) in s.ev:
    if t==0 and b==0:
        A.random(seed); slot=A.note_on(0,int(a),bb); w=A.voice_download(slot)
        if n<4: print('lib p=%d:'%int(a), ' '.join('%08x'%x for x in w[:12]))
        n+=1
