cd tests/cpp && g++ -std=c++17 -O2 -I ../../include local_fuser_drive.cpp -L ../../randt-slam_amd -lrandt_hip -Wl,-rpath,$PWD/../../randt-slam_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -o /tmp/lfd && cd ../.. && python - <<'P'
import numpy as np, sys, subprocess
sys.path.insert(0,'.')
from randt_slam_amd import synth
n_scans, per_lap = 300, 160
world = synth.make_world()
th = 2*np.pi*np.arange(n_scans)/per_lap
truth = np.stack([5.0*np.cos(th), 5.0*np.sin(th), th+np.pi/2], 1)
scans = np.ascontiguousarray(np.stack([synth.make_scan(world, truth[i], 71000+i) for i in range(n_scans)]), dtype=np.float32)
with open('/tmp/scans.bin','wb') as f:
    f.write(np.array([scans.shape[0], scans.shape[1]], dtype=np.int32).tobytes()); f.write(scans.tobytes())
for every in (1, 12, 40):
    for extra in ([], ['--loop-group','1'], ['--loop-group','4']):
        for rep in range(2):
            r = subprocess.run(['/tmp/lfd','/tmp/scans.bin','/tmp/p.txt','40','10','--slam','/tmp/g.txt','--loop-every',str(every)]+extra, capture_output=True, text=True)
        print(every, extra, [l for l in r.stdout.splitlines() if l.startswith('loop search') or l.startswith('batched')])
P
