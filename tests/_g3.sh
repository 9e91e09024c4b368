cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, torch, os, sys
sys.path.insert(0, '.')
from maskunet_amd import ops
from oracle import cv2_resize_oracle as R
z = np.load('tests/golden/resize_cases.npz')
for i in range(8):
    img = z[f'c{i}_img']; dw, dh = (int(v) for v in z[f'c{i}_dsize'])
    y, u8 = ops.resize_u8_to_nhwc(torch.from_numpy(img)[None].cuda(), (dw, dh), torch.float32, return_u8=True)
    want = z[f'c{i}_lin']
    got = u8[0].cpu().numpy()
    nb = int((got != want).sum())
    act = torch.from_numpy(want).float() / 255.0
    na = int((y[0, :, :, :3].cpu() != act).sum())
    print(i, img.shape, (dh, dw), 'byte mismatches', nb, 'of', want.size, 'act mismatches', na, 'pad', float(y[..., 3:].abs().max()))
    if nb:
        idx = np.argwhere(got != want)[:5]
        for a in idx: print('   at', a, 'got', got[tuple(a)], 'want', want[tuple(a)])
    if na and not nb:
        d = (y[0, :, :, :3].cpu() != act)
        j = d.nonzero()[:3]
        for a in j: print('   act at', a.tolist(), float(y[0][tuple(a)]), float(act[tuple(a)]), int(want[tuple(a.tolist())]))
PY
python tests/debug_noise.py 2 > gpurun_out/r03c_noise_links.txt 2>&1; tail -80 gpurun_out/r03c_noise_links.txt
