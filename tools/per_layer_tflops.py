"""Per-layer TFLOP/s of the implicit-GEMM conv kernels from a rocprofv3 kernel trace of ONE un-captured step
(`rocprofv3 --kernel-trace --output-format csv -- python bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline
--no-roofline`): the igemm launches between the last two Adam kernels are matched, in launch order, with the layer
sequence of a FlowNetC step (forward, then backward in reverse order: filter gradient before data gradient) and their
algorithmic GFLOP at B=4 384x512.

    python tools/per_layer_tflops.py <kernel_trace.csv>  >  profiles/rNN_per_layer.txt
"""
import csv, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
# expected launch order of MFMA kernels in a step with GFLOP (N=8, 384x512)
fwd = [('conv1',7.40),('conv2',40.27),('conv3',40.27),('conv_redir',0.40),('conv3_1',53.57),('conv4',14.50),('conv4_1',28.99),('conv5',7.25),('conv5_1',7.25),('conv6',3.62),('conv6_1',7.25),('deconv5',6.44),('deconv4',12.91),('deconv3',19.38),('deconv2',19.43)]
bwd = [('deconv2 wgrad',19.43),('deconv2 dgrad',19.43),('deconv3 wgrad',19.38),('deconv3 dgrad',19.38),('deconv4 wgrad',12.91),('deconv4 dgrad',12.91),('deconv5 wgrad',6.44),('deconv5 dgrad',6.44),
       ('conv6_1 wgrad',7.25),('conv6_1 dgrad',7.25),('conv6 wgrad',3.62),('conv6 dgrad',3.62),('conv5_1 wgrad',7.25),('conv5_1 dgrad',7.25),('conv5 wgrad',7.25),('conv5 dgrad',7.25),
       ('conv4_1 wgrad',28.99),('conv4_1 dgrad',28.99),('conv4 wgrad',14.5),('conv4 dgrad',14.5),('conv3_1 wgrad',53.57),('conv3_1 dgrad',53.57),('conv_redir wgrad',0.4),   # its dgrad runs in pointwise32_dgrad_kernel, not an igemm launch
       
       ('conv3 wgrad',40.27),('conv3 dgrad',40.27),('conv2 wgrad',40.27),('conv2 dgrad',40.27),('conv1 wgrad',7.4)]
seq = fwd + bwd
k = 0
tot = 0; totg = 0
for r in rows[a+1:b+1]:
    nm = r['Kernel_Name']
    if 'igemm_' not in nm: continue
    d = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    name, gf = seq[k] if k < len(seq) else ('?', 0); k += 1
    cfg = nm.split('<')[1].split('>')[0]
    bx = int(r['Grid_Size_X'])//256
    print('%-18s %-24s grid(%5d,%3s,%3s) %8.1f us  %6.1f TF' % (name, cfg, bx, r['Grid_Size_Y'], r['Grid_Size_Z'], d, gf/d*1e3))
    tot += d; totg += gf
print('total igemm %.1f us  %.1f GF -> %.1f TF' % (tot, totg, totg/tot*1e3))
