"""Oracle: DenseEnc (BaseBEVBackboneV2), CMA (ConvNeXt + DCNv2 hourglass), AFD / PFD distillation losses.
SURVEY 8(a) rows A6, A7, A8, A10.  Test infrastructure.

Follows
  pcdet/models/backbones_2d/base_bev_backbone.py:205-308          (BaseBEVBackboneV2)
  pcdet/models/backbones_2d/radar_distill_final.py:29-217         (Radar_Distill)
  pcdet/ops/basicblock/modules/Basicblock_convn.py:10-95          (ConvNeXtBlock, LayerNorm, GRN)
  pcdet/ops/basicblock/modules/modulated_deform_conv.py:14-64     (ModulatedDeformConv)
  pcdet/ops/basicblock/src/cuda/modulated_deform_im2col_cuda.cuh:24-194 (bilinear sampling, im2col)
  pcdet/ops/basicblock/src/cuda/modulated_deform_conv_cuda.cu:75-121    (addmm + bias)
"""
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- DCNv2
def modulated_deform_conv(x, offset, mask, weight, bias, stride=2, pad=1, dil=1):
    """DCNv2 forward, deformable_groups = groups = 1, as differentiable torch ops.

    Sampling position of tap t = i*kw + j at output (ho, wo):
        h = ho*stride - pad + i*dil + offset[:, 2t], w = wo*stride - pad + j*dil + offset[:, 2t+1]
    (modulated_deform_im2col_cuda.cuh:171-178).  The sample is taken only if -1 < h < H and
    -1 < w < W (:180); corners outside the map read as 0 (mdmcn_im2col_bilinear :24-54).
    columns = val * mask (:190); output = columns^T @ W^T + bias (modulated_deform_conv_cuda.cu:106-114).
    The bias is ALWAYS added (even when the module was built with bias=False; the parameter still
    exists, modulated_deform_conv.py:36-49).
    """
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    ho = torch.arange(Ho, dtype=x.dtype).view(1, Ho, 1)
    wo = torch.arange(Wo, dtype=x.dtype).view(1, 1, Wo)
    xf = x.reshape(B, C, H * W)
    cols = []
    for i in range(kh):
        for j in range(kw):
            t = i * kw + j
            h = ho * stride - pad + i * dil + offset[:, 2 * t]
            w = wo * stride - pad + j * dil + offset[:, 2 * t + 1]
            inside = (h > -1) & (w > -1) & (h < H) & (w < W)
            h_low = torch.floor(h)
            w_low = torch.floor(w)
            lh, lw = h - h_low, w - w_low
            hh, hw = 1 - lh, 1 - lw
            h_low, w_low = h_low.long(), w_low.long()
            h_high, w_high = h_low + 1, w_low + 1

            def corner(hi, wi, ok):
                ok = ok & inside
                flat = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
                v = torch.gather(xf, 2, flat).view(B, C, Ho, Wo)
                return v * ok.unsqueeze(1).to(x.dtype)

            v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
            v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
            v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
            v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
            val = (hh * hw).unsqueeze(1) * v1 + (hh * lw).unsqueeze(1) * v2 + \
                  (lh * hw).unsqueeze(1) * v3 + (lh * lw).unsqueeze(1) * v4
            cols.append(val * mask[:, t].unsqueeze(1))
    col = torch.stack(cols, dim=2)                      # (B, C, 9, Ho, Wo): column row index = c*9 + t
    col = col.reshape(B, C * kh * kw, Ho * Wo)
    out = torch.einsum("ok,bkn->bon", weight.reshape(Cout, -1), col) + bias.view(1, -1, 1)
    return out.view(B, Cout, Ho, Wo)


# ----------------------------------------------------------------------------- ConvNeXt block
def convnext_block(x, state, prefix, downsample):
    """ConvNeXtBlock.forward (Basicblock_convn.py:38-56)."""
    if downsample:
        om = F.conv2d(x, state[prefix + "conv_offset_mask1.weight"], state[prefix + "conv_offset_mask1.bias"],
                      stride=2, padding=1)
        o1, o2, m = torch.chunk(om, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        x = modulated_deform_conv(x, offset, torch.sigmoid(m), state[prefix + "down_layer.weight"],
                                  state[prefix + "down_layer.bias"], stride=2, pad=1)
    identity = x
    C = x.shape[1]
    x = F.conv2d(x, state[prefix + "dwconv.weight"], state[prefix + "dwconv.bias"], padding=3, groups=C)
    x = x.permute(0, 2, 3, 1)
    x = F.layer_norm(x, (C,), state[prefix + "norm.weight"], state[prefix + "norm.bias"], 1e-6)
    x = F.linear(x, state[prefix + "pwconv1.weight"], state[prefix + "pwconv1.bias"])
    x = F.gelu(x)
    # GRN (Basicblock_convn.py:84-95)
    gx = torch.norm(x, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    x = state[prefix + "grn.gamma"] * (x * nx) + state[prefix + "grn.beta"] + x
    x = F.linear(x, state[prefix + "pwconv2.weight"], state[prefix + "pwconv2.bias"])
    x = x.permute(0, 3, 1, 2)
    return x + identity


def _bn(x, state, prefix, training, eps, momentum):
    return F.batch_norm(x, state[prefix + "running_mean"], state[prefix + "running_var"],
                        state[prefix + "weight"], state[prefix + "bias"], training=training,
                        momentum=momentum, eps=eps)


# ----------------------------------------------------------------------------- DenseEnc
def dense_enc_block(x, state, prefix, n_layers, training):
    """One `blocks[idx]` Sequential of BaseBEVBackboneV2 (base_bev_backbone.py:222-249):
    ZeroPad2d(1)+Conv3x3(p0)+BN(eps1e-3,mom0.01)+ReLU then n_layers x [Conv3x3 p1 + BN + ReLU].
    Sequential indices: conv at 1,4,7,..., BN at 2,5,8,..."""
    for k in range(n_layers + 1):
        x = F.conv2d(x, state[f"{prefix}{1 + 3 * k}.weight"], None, padding=1)
        x = F.relu(_bn(x, state, f"{prefix}{2 + 3 * k}.", training, 1e-3, 0.01))
    return x


def dense_enc(x_conv4, x_conv5, state, prefix, training, layer_nums=(5, 5)):
    """BaseBEVBackboneV2.forward (base_bev_backbone.py:285-308).  deblocks was sliced [1:] (:282), so
    deblocks.0 is ConvTranspose2d(256, 256, 2, stride 2, bias=False)+BN+ReLU."""
    x = dense_enc_block(x_conv5, state, prefix + "blocks.1.", layer_nums[1], training)
    up = F.conv_transpose2d(x, state[prefix + "deblocks.0.0.weight"], None, stride=2)
    up = F.relu(_bn(up, state, prefix + "deblocks.0.1.", training, 1e-3, 0.01))
    x = torch.cat([x_conv4, up], dim=1)
    x = dense_enc_block(x, state, prefix + "blocks.0.", layer_nums[0], training)
    return up, x      # (spatial_features_2d_8x, spatial_features_2d)


# ----------------------------------------------------------------------------- CMA + Radar_Distill.forward
def radar_distill_forward(x_conv4, x_conv5, state, prefix, training):
    """Radar_Distill.forward (radar_distill_final.py:177-217).  decoder = ConvT(256,256,4,2,1)+BN(eps 1e-5,
    momentum 0.1: nn.BatchNorm2d defaults)+GELU; agg = Conv1x1(512->256)+BN+GELU."""
    def enc(x, name):
        x = convnext_block(x, state, f"{prefix}{name}.0.", True)
        return convnext_block(x, state, f"{prefix}{name}.1.", False)

    def dec(x, name):
        x = F.conv_transpose2d(x, state[f"{prefix}{name}.0.weight"], state[f"{prefix}{name}.0.bias"], stride=2, padding=1)
        return F.gelu(_bn(x, state, f"{prefix}{name}.1.", training, 1e-5, 0.1))

    def agg(x, name):
        x = F.conv2d(x, state[f"{prefix}{name}.0.weight"], state[f"{prefix}{name}.0.bias"])
        return F.gelu(_bn(x, state, f"{prefix}{name}.1.", training, 1e-5, 0.1))

    en16 = enc(x_conv4, "encoder_1")
    de8 = agg(torch.cat((dec(en16, "decoder_1"), x_conv4), dim=1), "agg_1")
    en32 = enc(en16, "encoder_2")
    de16 = agg(torch.cat((dec(en32, "decoder_2"), enc(de8, "encoder_3")), dim=1), "agg_2")
    x4 = agg(torch.cat((dec(de16, "decoder_3"), de8), dim=1), "agg_3")
    up, feat = dense_enc(x4, x_conv5, state, prefix, training)
    return {"radar_spatial_features_8x_2": x4, "radar_spatial_features_8x_1": de8,
            "radar_spatial_features_2d_8x": up, "radar_spatial_features_2d": feat}


# ----------------------------------------------------------------------------- AFD / PFD
def clip_sigmoid(x, eps=1e-4):
    """radar_distill_final.py:12-26."""
    return torch.clamp(torch.sigmoid(x), min=eps, max=1 - eps)


def low_loss(lidar_bev, radar_bev):
    """AFD: Radar_Distill.low_loss (radar_distill_final.py:82-109).  NaN when no inactive-radar /
    active-lidar cell exists (0/0), as in the reference."""
    B = radar_bev.shape[0]
    lidar_mask = (lidar_bev.sum(1).unsqueeze(1) > 0).float()
    radar_mask = radar_bev.sum(1).unsqueeze(1)
    act = (radar_mask > 0).float() + lidar_mask * 0.5
    m_ar = (act == 1.5).float()
    m_ir = (act == 1.0).float()
    m_ir = m_ir * (m_ar.sum() / m_ir.sum())
    mse = F.mse_loss(radar_bev, lidar_bev, reduction="none")
    l_ar = torch.sum(mse * m_ar) / B
    l_ir = torch.sum(mse * m_ir) / B
    feature_loss = 3e-4 * l_ar + 5e-5 * l_ir
    mask_loss = F.l1_loss(torch.sigmoid(radar_mask), lidar_mask)
    return feature_loss, mask_loss


def high_loss(radar_bev, radar_bev2, lidar_bev, lidar_bev2, heatmaps, radar_hm_logits):
    """PFD: Radar_Distill.high_loss (radar_distill_final.py:111-141)."""
    gt = torch.max(torch.cat(heatmaps, dim=1), dim=1, keepdim=True)[0]
    pr = torch.max(torch.cat([clip_sigmoid(h) for h in radar_hm_logits], dim=1), dim=1, keepdim=True)[0]
    fp = (gt < 0.1) & (pr > 0.1)
    fn = (gt > 0.1) & (pr < 0.1)
    tp = (gt > 0.1) & (pr > 0.1)
    w = torch.zeros_like(pr)
    w[tp + fn] = 5 / (tp + fn).sum()
    w[fp] = 1 / fp.sum()
    l1 = (F.l1_loss(radar_bev.softmax(1), lidar_bev.softmax(1), reduction="none") * w).sum()
    l2 = (F.l1_loss(radar_bev2.softmax(1), lidar_bev2.softmax(1), reduction="none") * w).sum()
    return 0.5 * (l1 + l2)


def distill_loss(lidar_x_conv4, radar_out, lidar_2d, lidar_2d_8x, heatmaps, radar_hm_logits):
    """Radar_Distill.get_loss (radar_distill_final.py:144-175): 5*low + 25*high."""
    f, m = low_loss(lidar_x_conv4, radar_out["radar_spatial_features_8x_2"])
    f8, m8 = low_loss(lidar_x_conv4, radar_out["radar_spatial_features_8x_1"])
    high = 25 * high_loss(radar_out["radar_spatial_features_2d"], radar_out["radar_spatial_features_2d_8x"],
                          lidar_2d, lidar_2d_8x, heatmaps, radar_hm_logits)
    low = 5 * (0.5 * (f + f8) + 0.5 * (m + m8))
    tb = {"low_feature_loss": low, "high_distill_loss": high, "distll_loss": low + high,
          "low_distill_de_8x_loss": f8, "low_distill_loss": f, "mask_loss": m, "mask_de_8x_loss": m8}
    return low + high, tb
