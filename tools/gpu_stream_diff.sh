#!/bin/bash
# developer helper (one gpurun call): decode a stream with the reference decoder and with the drop-in on the GPU back-end, say where the outputs differ first
#   tools/gpu_stream_diff.sh <stream.bit> <width> <height> <8|10> [400]
S=$1; W=$2; H=$3; BD=$4; CF=${5:-420}
R=oracle/_ref
export LD_LIBRARY_PATH=$R:$PWD/vvdec_amd:$LD_LIBRARY_PATH
$R/vvdecapp_ref -b $S -t 1 -v 1 -o /tmp/ref.yuv > /dev/null 2>&1
LD_PRELOAD=$PWD/vvdec_amd/libvvdec_amd.so $R/vvdecapp_dropin -b $S -t 1 -v 1 -o /tmp/gpu.yuv 2>&1 | tail -2
python - $W $H $BD $CF <<'PY'
import sys, numpy as np
W, H, bd, cf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dt = np.uint16 if bd > 8 else np.uint8
a, b = np.fromfile('/tmp/ref.yuv', dt), np.fromfile('/tmp/gpu.yuv', dt)
print('samples', len(a), len(b))
fs = W * H * 3 // 2 if cf == '420' else W * H
for p in range(min(len(a), len(b)) // fs):
    A, B = a[p * fs:(p + 1) * fs], b[p * fs:(p + 1) * fs]
    planes = [(A[:W * H].reshape(H, W), B[:W * H].reshape(H, W))]
    if cf == '420':
        cw, ch = W // 2, H // 2
        planes += [(A[W * H + k * cw * ch:W * H + (k + 1) * cw * ch].reshape(ch, cw), B[W * H + k * cw * ch:W * H + (k + 1) * cw * ch].reshape(ch, cw)) for k in range(2)]
    for c, (x, y) in enumerate(planes):
        d = np.argwhere(x != y)
        if len(d):
            print('picture %d (output order) plane %d: %d samples differ, bounding box x %d..%d y %d..%d, first at (x %d, y %d): ref %d gpu %d' % (p, c, len(d), d[:, 1].min(), d[:, 1].max(), d[:, 0].min(), d[:, 0].max(), d[0][1], d[0][0], x[d[0][0], d[0][1]], y[d[0][0], d[0][1]]))
            cells = sorted(set((int(r[1]) // 8 * 8, int(r[0]) // 8 * 8) for r in d))[:12]
            print('   8x8 cells touched:', cells)
PY
