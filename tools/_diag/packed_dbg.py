import os, sys, torch
sys.path.insert(0, "/root/repo")
from doubleattentionspeakerverification_amd.CNNs import VGG4L
from doubleattentionspeakerverification_amd import conv as hc
DEV = "cuda:0"
torch.manual_seed(7)
fe = VGG4L(1024).to(DEV).eval()
mel = torch.randn(2, 61, 80, device=DEV) * 2.0
def d(a, b): return float((a - b).abs().max()), float(b.abs().max())
with torch.no_grad():
    h11p = hc.conv3x3_c1_bias_relu(mel, fe.conv11.weight, fe.conv11.bias, pack=True)
    h11 = hc.conv3x3_c1_bias_relu(mel, fe.conv11.weight, fe.conv11.bias)
    y_p = hc.conv3x3_x8(h11p, fe.conv12.weight, fe.conv12.bias, None, x_packed=True)
    y_u = hc.conv3x3_x8(h11, fe.conv12.weight, fe.conv12.bias, None, x_packed=False)
    y_f = hc.conv3x3_bias_relu(h11, fe.conv12.weight, fe.conv12.bias, None, "split8")
    print("conv12: x8(unpacked) vs bias_relu(split8)", d(y_u, y_f), " packed-in vs fp32-in", d(y_p, y_f))
    # decode the packed conv11 output on the host and compare with fp32
    raw = h11p.permute(0, 2, 3, 1).contiguous().view(-1, 128 // 64, 64).cpu()     # [pix][slab][64 floats]
    b = raw.view(torch.uint8).view(-1, 2, 256)
    hi = b[:, :, :128].contiguous().view(torch.float16).float()                      # [pix][slab][64]
    ref = h11.permute(0, 2, 3, 1).contiguous().view(-1, 2, 64).cpu()
    print("conv11 packed hi vs fp32:", d(hi, ref), " hi == fp16(ref):", bool(torch.equal(hi, ref.half().float())))
    pp, pf = hc.maxpool2x2_ceil(y_f, pack=True), hc.maxpool2x2_ceil(y_f)
    rawp = pp.permute(0, 2, 3, 1).contiguous().view(-1, 2, 64).cpu().view(torch.uint8).view(-1, 2, 256)
    hip_ = rawp[:, :, :128].contiguous().view(torch.float16).float()
    refp = pf.permute(0, 2, 3, 1).contiguous().view(-1, 2, 64).cpu()
    print("pool packed hi == fp16(pool fp32):", bool(torch.equal(hip_, refp.half().float())))
    c21p = hc.conv3x3_x8(pp, fe.conv21.weight, fe.conv21.bias, None, x_packed=True, y_packed=True)
    c21f = hc.conv3x3_bias_relu(pf, fe.conv21.weight, fe.conv21.bias, None, "split8")
    rawc = c21p.permute(0, 2, 3, 1).contiguous().view(-1, 4, 64).cpu().view(torch.uint8).view(-1, 4, 256)
    hic = rawc[:, :, :128].contiguous().view(torch.float16).float()
    refc = c21f.permute(0, 2, 3, 1).contiguous().view(-1, 4, 64).cpu()
    print("conv21 packed-out hi == fp16(conv21 fp32):", bool(torch.equal(hic, refc.half().float())), d(hic, refc))
