export TMPDIR=/tmp
H="--odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-roofline-sections"
mkdir -p gpurun_out/sat
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES -d gpurun_out/sat/a -o run --output-format csv -- python bench.py $H --batch-scale 8 --streams 1 --steps 12 --warmup 2 --min-seconds 0 > /dev/null 2>gpurun_out/sat/a.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_IFETCH -d gpurun_out/sat/b -o run --output-format csv -- python bench.py $H --batch-scale 8 --streams 1 --steps 12 --warmup 2 --min-seconds 0 > /dev/null 2>gpurun_out/sat/b.err
python - <<'PY'
import csv,re,collections,glob
for d in ('a','b'):
    t=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob('gpurun_out/sat/%s/*counter_collection.csv'%d)[0])):
        m=re.search(r"(k_[a-z_0-9]+)",r['Kernel_Name'])
        if not m or int(r['Grid_Size'])<200000: continue
        t[(m.group(1),r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
        t[(m.group(1),r['Grid_Size'])]['dur'].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    for k,v in t.items():
        print(k,{c:round(sum(x)/len(x)) for c,x in v.items()})
PY
