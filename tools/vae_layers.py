"""Dev tool: per-layer table (us, TFLOP/s) of the last VAE3DLarge decode + encode pass in a rocprofv3 rocpd database made
from tools/one_vae_decode.py (B = 32).  Layers are matched by launch order; FLOP per sample from the layer shapes."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x from kernels order by start"))
B = 32
def conv(vox, cout, cin, taps): return 2.0 * vox * cout * cin * taps
# (label, flop per sample, number of launches it owns incl. split-K finish)
DEC = [("decoder_input", 2 * 256 * 32768, None), ("dec.0 convT 512>256 4>8", conv(512, 256, 512, 8), None),
       ("dec.2 res conv1 256 @8", conv(512, 256, 256, 27), None), ("dec.2 res conv2", conv(512, 256, 256, 27), None),
       ("dec.3 convT 256>128 8>16", conv(4096, 128, 256, 8), None),
       ("dec.5 res conv1 128 @16", conv(4096, 128, 128, 27), None), ("dec.5 res conv2", conv(4096, 128, 128, 27), None),
       ("dec.6 convT 128>64 16>32", conv(32768, 64, 128, 8), None),
       ("dec.8 res conv1 64 @32", conv(32768, 64, 64, 27), None), ("dec.8 res conv2", conv(32768, 64, 64, 27), None),
       ("dec.9 conv 64>32", conv(32768, 32, 64, 27), None),
       ("dec.11 res conv1 32", conv(32768, 32, 32, 27), None), ("dec.11 res conv2", conv(32768, 32, 32, 27), None),
       ("dec.12 conv 32>1 + sigmoid", conv(32768, 1, 32, 27), None)]
ENC = [("enc.0 conv 1>32", conv(32768, 32, 1, 27), None), ("enc.2 res conv1 32>64", conv(32768, 64, 32, 27), None),
       ("enc.2 shortcut 1x1", conv(32768, 64, 32, 1), None), ("enc.2 res conv2 64", conv(32768, 64, 64, 27), None),
       ("enc.3 conv k4s2 64 32>16", conv(4096, 64, 64, 64), None), ("enc.5 res conv1 64>128", conv(4096, 128, 64, 27), None),
       ("enc.5 shortcut", conv(4096, 128, 64, 1), None), ("enc.5 res conv2 128", conv(4096, 128, 128, 27), None),
       ("enc.6 conv k4s2 128 16>8", conv(512, 128, 128, 64), None), ("enc.8 res conv1 128>256", conv(512, 256, 128, 27), None),
       ("enc.8 shortcut", conv(512, 256, 128, 1), None), ("enc.8 res conv2 256", conv(512, 256, 256, 27), None),
       ("enc.9 conv k4s2 256 8>4", conv(64, 256, 256, 64), None), ("enc.11 res conv1 256>512", conv(64, 512, 256, 27), None),
       ("enc.11 shortcut", conv(64, 512, 256, 1), None), ("enc.11 res conv2 512", conv(64, 512, 512, 27), None),
       ("enc.12 conv k4 512 4>1", conv(1, 512, 512, 64), None), ("fc_mu | fc_logvar", 2 * 512 * 512, None)]
# group launches: a conv3d_finish_kernel belongs to the launch before it
def passes(rows):
    groups = []
    for n, s, e, *_ in rows:
        if ("conv3d_finish" in n or "skinny_finish" in n) and groups: groups[-1][1] += (e - s) / 1000; groups[-1][2] += 1
        elif "elementwise" in n or "f32_to_f16" in n: continue
        else: groups.append([n, (e - s) / 1000, 1])
    return groups
g = passes(rows)
if not any("pw_1x1" in x[0] for x in g):
    # shortcuts fused into conv2 (second K source): no shortcut launches; their FLOPs belong to conv2
    sc = {lab.split()[0]: fl for lab, fl, _ in ENC if "shortcut" in lab}
    ENC = [(lab + (" + shortcut" if "conv2" in lab and lab.split()[0] in sc else ""), fl + (sc.get(lab.split()[0], 0) if "conv2" in lab else 0), x)
           for lab, fl, x in ENC if "shortcut" not in lab]
# last decode starts at the last 'gemm_f16_kernel<128, 128, 2, 2, 2, 0' (decoder_input); encode starts at conv3d_first
names = [x[0] for x in g]
di = max(i for i, n in enumerate(names) if "gemm_f16_kernel" in n and ", 0, 64>" in n)
ei = max(i for i, n in enumerate(names) if "first" in n)
def table(title, layers, start):
    tot_t = tot_f = 0
    print(title)
    for k, (lab, fl, _) in enumerate(layers):
        n, us, cnt = g[start + k]
        tot_t += us; tot_f += fl * B
        print(f"  {lab:30s} {us:8.1f} us  {fl * B / us / 1e6:7.0f} TF/s   {n[:50]}{' +finish' if cnt > 1 else ''}")
    print(f"  {'total':30s} {tot_t:8.1f} us  {tot_f / tot_t / 1e6:7.0f} TF/s = {tot_f / tot_t / 1e6 / 2500 * 100:.1f} % of 2.5 PF")
if di < ei: table("decode (B=32)", DEC, di); table("encode (B=32)", ENC, ei)
else:
    table("encode (B=32)", ENC, ei); table("decode (B=32)", DEC, di)
