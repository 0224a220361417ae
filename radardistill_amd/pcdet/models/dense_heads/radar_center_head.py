"""CenterHead (teacher) / Radar_CenterHead (student) on the MI355X kernels.

Constructor signature, module tree (shared_conv, heads_list.H.{center,center_z,dim,rot,vel,iou,hm}), forward_ret_dict,
batch_dict keys, target format and loss values follow the reference's
pcdet/models/dense_heads/radar_center_head.py:28-440 and center_head.py:390-424.
Convolutions run on the implicit-GEMM MFMA kernel; the loss has no host synchronisation (see utils/loss_utils.py).
Only the training path of the distill config is implemented: `generate_predicted_boxes` (eval decode + NMS) is a
"next" row of SURVEY section 8(f).
"""
import copy

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import kaiming_normal_

from radardistill_amd import autograd as A
from radardistill_amd import dense as D
from ...utils import loss_utils
from ...utils.loss_utils import IouLoss, IouRegLoss
from ..model_utils import centernet_utils


class SeparateHead(nn.Module):
    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for cur_name in self.sep_head_dict:
            output_channels = self.sep_head_dict[cur_name]['out_channels']
            num_conv = self.sep_head_dict[cur_name]['num_conv']
            fc_list = []
            for k in range(num_conv - 1):
                fc_list.append(nn.Sequential(
                    nn.Conv2d(input_channels, input_channels, kernel_size=3, stride=1, padding=1, bias=use_bias),
                    nn.BatchNorm2d(input_channels), nn.ReLU()))
            fc_list.append(nn.Conv2d(input_channels, output_channels, kernel_size=3, stride=1, padding=1, bias=True))
            fc = nn.Sequential(*fc_list)
            if 'hm' in cur_name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        kaiming_normal_(m.weight.data)
                        if hasattr(m, "bias") and m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            self.__setattr__(cur_name, fc)

    def forward(self, x):
        """x: (rows, B, H, W) channels-last rows of the shared feature."""
        ret_dict = {}
        for cur_name in self.sep_head_dict:
            fc = self.__getattr__(cur_name)
            state = x
            for m in fc:
                if isinstance(m, nn.Sequential):
                    out, B, H, W = D.conv_bn_act(None, m[0], m[1], None, act=1, return_rows=True, in_rows=state)
                else:
                    out, B, H, W = D.conv_bn_act(None, m, None, None, act=0, return_rows=True, in_rows=state)
                state = (out, B, H, W)
            ret_dict[cur_name] = A.rows_to_nchw(*state)
        return ret_dict


class _BranchBatch:
    """Execution plan that runs every branch of every SeparateHead of one CenterHead together (MI355X-first; the module tree and
    state_dict stay the reference's): the 42 first convolutions (64->64) as ONE implicit-GEMM conv with Cout = 42*64 and ONE
    BatchNorm+ReLU over those channels, the 42 final convolutions (64 -> 1..3) as ONE vector-ALU launch (nconv.hip).
    Output columns are grouped [hm | center | center_z | dim | rot | vel | iou], heads inner, so the batched loss reads the
    stacked maps as views.  `build` returns None when a head does not have the standard two-stage 64-channel branches."""

    @staticmethod
    def build(head):
        from radardistill_amd import kernels as K
        names = list(head.heads_list[0].sep_head_dict.keys())
        group_names = (['hm'] if 'hm' in names else []) + [n for n in names if n != 'hm']
        branches, groups, col = [], {}, 0
        for name in group_names:
            c0 = col
            widths = []
            for h, sh in enumerate(head.heads_list):
                if list(sh.sep_head_dict.keys()) != names:
                    return None
                fc = getattr(sh, name)
                if len(fc) != 2 or not isinstance(fc[0], nn.Sequential) or not isinstance(fc[1], nn.Conv2d):
                    return None
                c1, bn, c2 = fc[0][0], fc[0][1], fc[1]
                ok = (isinstance(c1, nn.Conv2d) and isinstance(bn, (nn.BatchNorm2d, nn.SyncBatchNorm)) and c1.in_channels == 64 and c1.out_channels == 64
                      and c2.in_channels == 64 and 1 <= c2.out_channels <= 4 and c2.bias is not None
                      and all(c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1
                              for c in (c1, c2)))
                if not ok:
                    return None
                branches.append((h, name, c1, bn, c2, col))
                widths.append(c2.out_channels)
                col += c2.out_channels
            groups[name] = (c0, widths)
        if not branches or len(branches) > 64:
            return None
        bn0 = branches[0][3]
        if any(b[3].eps != bn0.eps or b[3].momentum != bn0.momentum or (b[2].bias is None) != (branches[0][2].bias is None) for b in branches):
            return None
        plan = _BranchBatch()
        plan.branches, plan.groups, plan.names, plan.n_heads, plan.no = branches, groups, names, len(head.heads_list), col
        plan.tab = K.BranchTable([64 * i for i in range(len(branches))], [b[5] for b in branches], [b[4].out_channels for b in branches])
        plan.frozen_cache = None
        return plan

    def _frozen_tensors(self, frag=False):
        """Teacher: concatenated kernel-layout weights and folded BatchNorm, rebuilt only when a source tensor changes.
        frag: the first-stage weights in fragment-major split format (kernels.wants_frag_weights)."""
        from radardistill_amd import kernels as K
        src = [t for (_, _, c1, bn, c2, _) in self.branches for t in (c1.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, c2.weight, c2.bias)]
        b3 = K.get_conv_math() == "bf16x3"
        ver = (tuple(t._version for t in src), src[0].data_ptr(), b3, bool(frag), A._FROZEN_EPOCH[0])
        if self.frozen_cache is None or self.frozen_cache[0] != ver:
            with torch.no_grad():
                w1 = torch.cat([b[2].weight for b in self.branches], 0).contiguous()
                # bf16x3: the operand is kept pre-split like every other frozen weight (the in-kernel split made this one launch 3x slower)
                w1k = K.weight_layout_split(w1, w1.shape[0], 64, 9, 1, frag=frag) if b3 else K.weight_layout(w1, w1.shape[0], 64, 9, 1)
                b1 = torch.cat([b[2].bias for b in self.branches]) if self.branches[0][2].bias is not None else None
                rstd = torch.rsqrt(torch.cat([b[3].running_var for b in self.branches]) + self.branches[0][3].eps)
                scale = (torch.cat([b[3].weight for b in self.branches]) * rstd).contiguous()
                shift = (torch.cat([b[3].bias for b in self.branches]) - torch.cat([b[3].running_mean for b in self.branches]) * scale).contiguous()
                w2 = torch.cat([b[4].weight for b in self.branches], 0).contiguous()
                b2 = torch.cat([b[4].bias for b in self.branches]).contiguous()
            self.frozen_cache = (ver, w1k, b1, scale, shift, w2, b2)
        return self.frozen_cache[1:]

    def _train_store(self):
        """Training: the branches' parameters as persistent concatenated leaves (autograd.ConcatLeaves), built once; None when a
        branch parameter is frozen (the per-branch path handles mixed cases)."""
        st = getattr(self, '_store', None)
        first = self.branches[0]
        if st is not None and st['w1'].leaves[0] is first[2].weight and st['w1'].cat.device == first[2].weight.device:
            return st if st['ok'] else None
        groups = {'w1': [b[2].weight for b in self.branches], 'gamma': [b[3].weight for b in self.branches],
                  'beta': [b[3].bias for b in self.branches], 'w2': [b[4].weight for b in self.branches], 'b2': [b[4].bias for b in self.branches]}
        if first[2].bias is not None:
            groups['b1'] = [b[2].bias for b in self.branches]
        st = {k: A.ConcatLeaves(v) for k, v in groups.items()}
        st['ok'] = all(c.valid() for c in st.values())
        self._store = st
        return st if st['ok'] else None

    def run(self, x):
        """x = (rows (B*H*W, 64), B, H, W) -> (out (B*H*W, NO), per-head dicts of NCHW views)."""
        from radardistill_amd import kernels as K
        rows, B, H, W = x
        C1 = 64 * len(self.branches)
        spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
        bns = [b[3] for b in self.branches]
        training = bns[0].training
        if any(bn.training != training for bn in bns):
            return None
        flags = getattr(self, '_frozen_flag', None)             # walking 126 modules' parameters every step cost 0.6 ms
        if flags is None or flags[0] is not training:
            flags = self._frozen_flag = (training, not any(p.requires_grad for b in self.branches for m in b[2:5] for p in m.parameters()))
        params_frozen = flags[1]
        if (not torch.is_grad_enabled() or (params_frozen and not rows.requires_grad)) and not training:
            frag = K.wants_frag_weights(spec.fwd_ix, rows.shape[0], rows.shape[0], 64, C1, 9)
            w1k, b1, scale, shift, w2, b2 = self._frozen_tensors(frag)
            y = K.conv_fwd(rows, w1k, 9, b1, rows.shape[0], C1, spec.fwd_ix, scale=scale, shift=shift, relu=True,
                           w_split=2 if frag else K.get_conv_math() == "bf16x3")
            out = K.nconv_fwd(y, w2, b2, B, H, W, self.tab)
        elif training:
            # parameters of the 42 branches concatenated: persistent leaves whose gradients are handed to the branch parameters as
            # views (autograd.ConcatLeaves); under HIP-graph capture (torch.autograd.grad over the module's own parameters) or with
            # partly frozen branches: torch.cat inside the graph
            st = None if (A.CAPTURING[0] or not A.CONCAT_LEAVES[0]) else self._train_store()

            def cat(key, mod, name):
                if st is not None:
                    return st[key].refresh()
                leaves = [getattr(b[mod], name) for b in self.branches]
                t = torch.cat(leaves, 0)
                t._rd_leaves = leaves          # who receives the gradient (autograd.param_grad_stream)
                return t

            w1 = cat('w1', 2, 'weight')
            b1 = cat('b1', 2, 'bias') if self.branches[0][2].bias is not None else None
            stats = A.zeros_stats(2 * C1, rows.device)
            raw = A.conv(rows, w1, b1, spec, C1, stats, bias_feeds_bn=True)
            gamma = cat('gamma', 3, 'weight')
            beta = cat('beta', 3, 'bias')
            with torch.no_grad():
                rm = torch.cat([bn.running_mean for bn in bns])
                rv = torch.cat([bn.running_var for bn in bns])
            # BatchNorm + ReLU + the 42 narrow convolutions as one autograd node (the gradient of the 352 MB activation tensor is
            # never written: autograd.bn_relu_nconv_train)
            out = A.bn_relu_nconv_train(raw, gamma, beta, rm, rv, float(bns[0].eps), float(bns[0].momentum), stats, bns,
                                        cat('w2', 4, 'weight'), cat('b2', 4, 'bias'), B, H, W, self.tab)
            with torch.no_grad():
                torch._foreach_copy_([bn.running_mean for bn in bns], list(rm.split(64)))
                torch._foreach_copy_([bn.running_var for bn in bns], list(rv.split(64)))
        else:
            return None                       # eval-mode BatchNorm with gradients: rare, the per-branch path handles it
        o4 = out.view(B, H, W, self.no)
        dicts = [dict() for _ in range(self.n_heads)]
        for (h, name, _, _, c2, col) in self.branches:
            dicts[h][name] = o4[..., col:col + c2.out_channels].permute(0, 3, 1, 2)
        dicts = [{n: d[n] for n in self.names} for d in dicts]       # the reference's key order
        return o4, dicts


class Radar_CenterHead(nn.Module):
    FEATURE_KEY = 'radar_spatial_features_2d'
    IS_TEACHER = False

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True, bn_folding=False):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        self.feature_map_stride = self.model_cfg.TARGET_ASSIGNER_CONFIG.get('FEATURE_MAP_STRIDE', None)
        self.class_names = class_names
        self.class_names_each_head = []
        self.class_id_mapping_each_head = []
        self.waymo = model_cfg.get("waymo", False)
        for cur_class_names in self.model_cfg.CLASS_NAMES_EACH_HEAD:
            self.class_names_each_head.append([x for x in cur_class_names if x in class_names])
            self.class_id_mapping_each_head.append(torch.from_numpy(np.array(
                [self.class_names.index(x) for x in cur_class_names if x in class_names])))
        total_classes = sum([len(x) for x in self.class_names_each_head])
        assert total_classes == len(self.class_names), f'class_names_each_head={self.class_names_each_head}'
        self.shared_conv = nn.Sequential(
            nn.Conv2d(input_channels, self.model_cfg.SHARED_CONV_CHANNEL, 3, stride=1, padding=1,
                      bias=self.model_cfg.get('USE_BIAS_BEFORE_NORM', False)),
            nn.BatchNorm2d(self.model_cfg.SHARED_CONV_CHANNEL), nn.ReLU())
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = self.model_cfg.SEPARATE_HEAD_CFG
        for idx, cur_class_names in enumerate(self.class_names_each_head):
            cur_head_dict = copy.deepcopy(dict(self.separate_head_cfg.HEAD_DICT))
            cur_head_dict['hm'] = dict(out_channels=len(cur_class_names), num_conv=self.model_cfg.NUM_HM_CONV)
            self.heads_list.append(SeparateHead(input_channels=self.model_cfg.SHARED_CONV_CHANNEL, sep_head_dict=cur_head_dict,
                                                init_bias=-2.19, use_bias=self.model_cfg.get('USE_BIAS_BEFORE_NORM', False)))
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}
        self.with_iou = 'iou' in self.model_cfg.SEPARATE_HEAD_CFG.HEAD_DICT
        self.with_iou_reg = self.model_cfg.get("IOU_REG", False)
        if self.with_iou:
            self.crit_iou = IouLoss()
        self.crit_iou_reg = None
        if self.with_iou_reg:
            self.crit_iou_reg = IouRegLoss(self.with_iou_reg)
        self.build_losses()
        self.bn_folding = bn_folding

    def build_losses(self):
        self.add_module('hm_loss_func', loss_utils.FocalLossCenterNet())
        self.add_module('reg_loss_func', loss_utils.RegLossCenterNet())

    # ------------------------------------------------------------------ target assignment (host, exact reference arithmetic)
    def assign_target_of_single_head(self, num_classes, gt_boxes, feature_map_size, feature_map_stride, num_max_objs=500,
                                     gaussian_overlap=0.1, min_radius=2):
        """gt_boxes: CPU (n, 10) with the last column the 1-based class id inside this head; feature_map_size = [x, y]."""
        fx, fy = int(feature_map_size[0]), int(feature_map_size[1])
        heatmap = gt_boxes.new_zeros(num_classes, fy, fx)
        ret_boxes = gt_boxes.new_zeros((num_max_objs, gt_boxes.shape[-1] - 1 + 1))
        gt_box = gt_boxes.new_zeros((num_max_objs, gt_boxes.shape[-1] - (1 if self.waymo else 3)))
        inds = gt_boxes.new_zeros(num_max_objs).long()
        mask = gt_boxes.new_zeros(num_max_objs).long()
        n = min(num_max_objs, gt_boxes.shape[0])
        if n == 0:
            return heatmap, ret_boxes, inds, mask, gt_box
        x, y, z = gt_boxes[:, 0], gt_boxes[:, 1], gt_boxes[:, 2]
        coord_x = (x - self.point_cloud_range[0]) / self.voxel_size[0] / feature_map_stride
        coord_y = (y - self.point_cloud_range[1]) / self.voxel_size[1] / feature_map_stride
        coord_x = torch.clamp(coord_x, min=0, max=fx - 0.5)
        coord_y = torch.clamp(coord_y, min=0, max=fy - 0.5)
        center = torch.cat((coord_x[:, None], coord_y[:, None]), dim=-1)
        center_int = center.int()
        dx = gt_boxes[:, 3] / self.voxel_size[0] / feature_map_stride
        dy = gt_boxes[:, 4] / self.voxel_size[1] / feature_map_stride
        radius = torch.clamp_min(centernet_utils.gaussian_radius(dx, dy, min_overlap=gaussian_overlap).int(), min=min_radius)
        ok = (dx[:n] > 0) & (dy[:n] > 0) & (center_int[:n, 0] >= 0) & (center_int[:n, 0] <= fx) & \
            (center_int[:n, 1] >= 0) & (center_int[:n, 1] <= fy)
        cls = (gt_boxes[:n, -1] - 1).long()
        rad = radius[:n].tolist()
        ci = center_int[:n].tolist()
        for k in torch.nonzero(ok).flatten().tolist():
            centernet_utils.draw_gaussian_to_heatmap(heatmap[cls[k]], ci[k], rad[k])
        # vectorised slot fill (rows that fail `ok` stay zero, like the reference's `continue`)
        okf = ok
        inds[:n] = torch.where(okf, center_int[:n, 1].long() * fx + center_int[:n, 0].long(), inds[:n])
        mask[:n] = okf.long()
        rb = torch.zeros((n, ret_boxes.shape[1]), dtype=gt_boxes.dtype)
        rb[:, 0:2] = center[:n] - center_int[:n].float()
        rb[:, 2] = z[:n]
        rb[:, 3:6] = gt_boxes[:n, 3:6].log()
        rb[:, 6] = torch.cos(gt_boxes[:n, 6])
        rb[:, 7] = torch.sin(gt_boxes[:n, 6])
        if gt_boxes.shape[1] > 8:
            rb[:, 8:] = gt_boxes[:n, 7:-1]
        ret_boxes[:n] = torch.where(okf[:, None], rb, ret_boxes[:n])
        gt_box[:n, :7] = torch.where(okf[:, None], gt_boxes[:n, :7], gt_box[:n, :7])
        return heatmap, ret_boxes, inds, mask, gt_box

    def _target_cfg(self, feature_map_size_xy):
        from radardistill_amd.native import TargetCfg
        cfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        c = TargetCfg()
        c.n_classes, c.n_heads = len(self.class_names), len(self.class_names_each_head)
        c.n_channels = sum(len(x) for x in self.class_names_each_head)
        off = 0
        for h, names in enumerate(self.class_names_each_head):
            c.chan_off[h] = off
            off += len(names)
            for l, n in enumerate(names):
                g = self.class_names.index(n) + 1
                c.head_of_class[g], c.local_of_class[g] = h, l
        c.pcr0, c.pcr1 = float(self.point_cloud_range[0]), float(self.point_cloud_range[1])
        c.vs0, c.vs1 = float(self.voxel_size[0]), float(self.voxel_size[1])
        c.stride, c.fx, c.fy = int(cfg.FEATURE_MAP_STRIDE), int(feature_map_size_xy[0]), int(feature_map_size_xy[1])
        c.max_objs, c.min_radius, c.overlap = int(cfg.NUM_MAX_OBJS), int(cfg.MIN_RADIUS), float(cfg.GAUSSIAN_OVERLAP)
        return c

    def assign_targets_gpu(self, gt_boxes, feature_map_size):
        """Same targets as assign_targets, produced by ONE HIP kernel from the device copy of gt_boxes (targets.hip)."""
        from radardistill_amd import kernels as K
        fm = list(feature_map_size)[::-1]
        stk = K.center_targets(gt_boxes.float().contiguous(), self._target_cfg(fm))
        ret_dict = {'heatmaps': [], 'target_boxes': [], 'inds': [], 'masks': [], 'heatmap_masks': [], 'gt_box': []}
        c0 = 0
        for h, names in enumerate(self.class_names_each_head):
            ret_dict['heatmaps'].append(stk['heatmaps'][:, c0:c0 + len(names)])
            c0 += len(names)
            ret_dict['target_boxes'].append(stk['target_boxes'][h]); ret_dict['inds'].append(stk['inds'][h])
            ret_dict['masks'].append(stk['masks'][h]); ret_dict['gt_box'].append(stk['gt_box'][h])
        ret_dict['_stacked'] = stk
        return ret_dict

    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """gt_boxes (B, M, 10).  Computed on the host like the reference (radar_center_head.py:189-252) but from a host copy
        of the boxes when the caller provides one (`gt_boxes_host`), so no device->host sync is needed; the result is
        uploaded once per head.  The reference's in-place rewrite of the class column is applied to a clone."""
        feature_map_size = feature_map_size[::-1]
        cfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        host = kwargs.get('gt_boxes_host', None)
        dev = gt_boxes.device
        gt_cpu = torch.as_tensor(host).float().clone() if host is not None else gt_boxes.detach().cpu().clone()
        batch_size = gt_cpu.shape[0]
        cls_all = gt_cpu[:, :, -1].long()
        name_to_global = {n: i + 1 for i, n in enumerate(self.class_names)}
        per_head = []
        for cur_class_names in self.class_names_each_head:
            ids = torch.tensor([name_to_global[n] for n in cur_class_names])
            lists = [[], [], [], [], []]
            for bs_idx in range(batch_size):
                cur = gt_cpu[bs_idx]
                c = cls_all[bs_idx]
                sel = (c[:, None] == ids[None, :])
                keep = sel.any(1)
                single = cur[keep].clone()
                single[:, -1] = (sel[keep].float().argmax(1) + 1).float()
                out = self.assign_target_of_single_head(
                    num_classes=len(cur_class_names), gt_boxes=single, feature_map_size=feature_map_size,
                    feature_map_stride=cfg.FEATURE_MAP_STRIDE, num_max_objs=cfg.NUM_MAX_OBJS,
                    gaussian_overlap=cfg.GAUSSIAN_OVERLAP, min_radius=cfg.MIN_RADIUS)
                for l, o in zip(lists, out):
                    l.append(o)
            per_head.append([torch.stack(l, dim=0) for l in lists])
        # one upload per quantity: heat-maps concatenated over heads on the channel axis, the slot tensors stacked over heads
        hm_all = torch.cat([ph[0] for ph in per_head], dim=1).to(dev, non_blocking=True)
        tb_all = torch.stack([ph[1] for ph in per_head], dim=0).to(dev, non_blocking=True)
        ind_all = torch.stack([ph[2] for ph in per_head], dim=0).to(dev, non_blocking=True)
        mask_all = torch.stack([ph[3] for ph in per_head], dim=0).to(dev, non_blocking=True)
        gb_all = torch.stack([ph[4] for ph in per_head], dim=0).to(dev, non_blocking=True)
        ret_dict = {'heatmaps': [], 'target_boxes': [], 'inds': [], 'masks': [], 'heatmap_masks': [], 'gt_box': []}
        c0 = 0
        for h, cur_class_names in enumerate(self.class_names_each_head):
            ret_dict['heatmaps'].append(hm_all[:, c0:c0 + len(cur_class_names)])
            c0 += len(cur_class_names)
            ret_dict['target_boxes'].append(tb_all[h]); ret_dict['inds'].append(ind_all[h])
            ret_dict['masks'].append(mask_all[h]); ret_dict['gt_box'].append(gb_all[h])
        ret_dict['_stacked'] = {'heatmaps': hm_all, 'target_boxes': tb_all, 'inds': ind_all, 'masks': mask_all, 'gt_box': gb_all}
        return ret_dict

    def sigmoid(self, x):
        return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)

    # ------------------------------------------------------------------ loss
    def get_loss(self):
        """All task heads in ONE batched pass (same per-head values and tb_dict entries as the reference's per-head loop,
        radar_center_head.py:258-330): heat-map channels concatenated, regression maps stacked over heads."""
        pred_dicts = self.forward_ret_dict['pred_dicts']
        target_dicts = self.forward_ret_dict['target_dicts']
        if '_stacked' not in target_dicts or not (self.with_iou and self.with_iou_reg):
            return self.get_loss_per_head()
        st = target_dicts['_stacked']
        lw = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        nh = len(pred_dicts)
        dev = st['heatmaps'].device
        tb_dict = {}
        fused = self._fused_loss(st, lw, nh)
        if fused is not None:
            loss, per_head = fused
            for idx in range(nh):
                tb_dict['hm_loss_head_%d' % idx] = per_head[idx, 0]
                tb_dict['loc_loss_head_%d' % idx] = per_head[idx, 1]
                tb_dict['iou_loss_head_%d' % idx] = per_head[idx, 2]
                tb_dict['iou_reg_loss_head_%d' % idx] = per_head[idx, 3]
            tb_dict['rpn_loss'] = loss.detach()
            return loss, tb_dict
        # ---- focal loss on all heat-map channels at once; per-head normalisation by that head's positives
        nc = [p['hm'].shape[1] for p in pred_dicts]
        head_of_ch = A.const_tensor(("head_of_ch", tuple(nc)), [h for h, c in enumerate(nc) for _ in range(c)], dev, torch.int64)
        # batched branches (head_forward): all maps live in ONE (B, H, W, NO) tensor, columns [hm | center | ... ], heads inner
        stacked = self.forward_ret_dict.get('pred_stacked', None)
        if stacked is not None and any(len(set(w)) != 1 for n, (_, w) in stacked[1].items() if n != 'hm'):
            stacked = None
        if stacked is not None:
            o4, groups = stacked
            hm = self.sigmoid(o4[..., groups['hm'][0]:groups['hm'][0] + sum(groups['hm'][1])].permute(0, 3, 1, 2))
        else:
            hm = self.sigmoid(torch.cat([p['hm'] for p in pred_dicts], dim=1))
        gt = st['heatmaps']
        pos_inds = gt.eq(1).float()
        neg_inds = gt.lt(1).float()
        pos_loss = (torch.log(hm) * torch.pow(1 - hm, 2) * pos_inds).sum((0, 2, 3))
        neg_loss = (torch.log(1 - hm) * torch.pow(hm, 2) * torch.pow(1 - gt, 4) * neg_inds).sum((0, 2, 3))
        per_ch = torch.stack([pos_loss + neg_loss, pos_inds.sum((0, 2, 3))], dim=0)                 # (2, n_ch)
        per_head = torch.zeros((2, nh), device=dev, dtype=per_ch.dtype).index_add_(1, head_of_ch, per_ch)
        hm_loss = -per_head[0] / torch.clamp_min(per_head[1], 1.0) * lw['cls_weight']             # (nh,)
        # ---- regression maps stacked over heads: (nh*B, c, H, W)
        B = pred_dicts[0]['center'].shape[0]

        def stk(name):
            if stacked is not None:
                c0, widths = groups[name]
                c = widths[0]
                Bq, Hq, Wq, _ = o4.shape
                return o4[..., c0:c0 + nh * c].reshape(Bq, Hq, Wq, nh, c).permute(3, 0, 4, 1, 2).reshape(nh * Bq, c, Hq, Wq)
            return torch.cat([p[name] for p in pred_dicts], dim=0)

        center, center_z, dim_, rot, vel, iou = [stk(n) for n in ('center', 'center_z', 'dim', 'rot', 'vel', 'iou')]
        inds = st['inds'].reshape(nh * B, -1)
        masks = st['masks'].reshape(nh * B, -1)
        K = inds.shape[1]
        mb = masks.bool()
        mf = masks.float()
        n_head = mf.view(nh, -1).sum(1)                                                              # positives per head
        pred_boxes = torch.cat([center, center_z, dim_, rot, vel], dim=1)                            # HEAD_ORDER minus iou
        pred = loss_utils._transpose_and_gather_feat(pred_boxes, inds)                               # (nh*B, K, 10)
        tgt = st['target_boxes'].reshape(nh * B, K, -1)
        m = mf.unsqueeze(2) * (~torch.isnan(tgt)).float()
        reg = torch.abs(pred * m - tgt * m).view(nh, B * K, -1).sum(1) / torch.clamp_min(n_head, 1.0).unsqueeze(1)     # (nh, 10)
        loc_loss = (reg * A.const_tensor("code_weights", list(lw['code_weights']), dev)).sum(1) * lw['loc_weight']     # (nh,)
        # ---- decode every cell to a box (parity trap kept: int() truncates the range origin, radar_center_head.py:309-310)
        batch_dim = torch.exp(torch.clamp(dim_, min=-5, max=5))
        batch_rot = torch.atan2(rot[:, 1:2], rot[:, 0:1])
        _, _, H, W = batch_dim.shape
        ys, xs = torch.meshgrid(torch.arange(0, H, device=dev), torch.arange(0, W, device=dev), indexing='ij')
        xs = xs.view(1, 1, H, W).to(batch_dim) + center[:, 0:1]
        ys = ys.view(1, 1, H, W).to(batch_dim) + center[:, 1:2]
        xs = xs * int(self.feature_map_stride) * self.voxel_size[0] + int(self.point_cloud_range[0])
        ys = ys * int(self.feature_map_stride) * self.voxel_size[1] + int(self.point_cloud_range[1])
        box_map = torch.cat([xs, ys, center_z, batch_dim, batch_rot], dim=1)                         # (nh*B, 7, H, W)
        pb = loss_utils._masked_boxes(loss_utils._transpose_and_gather_feat(box_map, inds), mb)      # (nh*B, K, 7)
        gtb = loss_utils._masked_boxes(st['gt_box'].reshape(nh * B, K, -1)[..., :7], mb)
        # IouLoss: L1(iou head, 2*IoU3D(detached pred, gt) - 1) / (n + 1e-4)
        iou_pred = loss_utils._transpose_and_gather_feat(iou, inds)                                  # (nh*B, K, 1)
        target = loss_utils.iou3d_nms_utils.boxes_aligned_iou3d_gpu(pb.detach().reshape(-1, 7), gtb.reshape(-1, 7)).view(nh * B, K, 1)
        iou_loss = (torch.abs(iou_pred - (2 * target - 1)) * mf.unsqueeze(-1)).view(nh, -1).sum(1) / (n_head + 1e-4)
        # IouRegLoss (DIoU): sum(1 - diou) / (n + 1e-4)
        diou = loss_utils.bbox3d_overlaps_diou(pb.reshape(-1, 7), gtb.reshape(-1, 7)).view(nh * B, K)
        iou_reg_loss = ((1.0 - diou) * mf).view(nh, -1).sum(1) / (n_head + 1e-4)
        per_head_total = hm_loss + loc_loss + iou_loss + lw['loc_weight'] * iou_reg_loss
        loss = per_head_total.sum().view(1)
        hm_d, loc_d, iou_d, reg_d = hm_loss.detach(), loc_loss.detach(), iou_loss.detach(), iou_reg_loss.detach()
        for idx in range(nh):
            tb_dict['hm_loss_head_%d' % idx] = hm_d[idx]
            tb_dict['loc_loss_head_%d' % idx] = loc_d[idx]
            tb_dict['iou_loss_head_%d' % idx] = iou_d[idx]
            tb_dict['iou_reg_loss_head_%d' % idx] = reg_d[idx]
        tb_dict['rpn_loss'] = loss.detach()
        return loss, tb_dict

    def _fused_loss(self, st, lw, nh):
        """The whole loss in 3 launches forward / 2 backward (centerloss.hip) when the branches were batched into one
        (B, H, W, NO) map; None -> the torch expressions below (same values; LOSS_CONFIG.FUSED: False forces them)."""
        stacked = self.forward_ret_dict.get('pred_stacked', None)
        if stacked is None or not self.model_cfg.LOSS_CONFIG.get('FUSED', True):
            return None
        o4, groups = stacked
        names = ('center', 'center_z', 'dim', 'rot', 'vel', 'iou')
        want = {'center': 2, 'center_z': 1, 'dim': 3, 'rot': 2, 'vel': 2, 'iou': 1}
        if not o4.is_cuda or any(n not in groups for n in names + ('hm',)) or any(list(groups[n][1]) != [want[n]] * nh for n in names):
            return None
        nc = list(groups['hm'][1])
        if nh > 8 or sum(nc) > 16 or len(lw['code_weights']) != 10 or st['target_boxes'].shape[-1] < 10:
            return None
        from radardistill_amd.native import CenterLossCfg
        B, H, W, NO = o4.shape
        key = (B, H, W, NO, tuple(nc), tuple((n, groups[n][0]) for n in names + ('hm',)), st['inds'].shape[2])
        cached = getattr(self, '_loss_cfg_cache', None)
        if cached is None or cached[0] != key:
            c = CenterLossCfg()
            c.B, c.H, c.W, c.NO, c.n_heads, c.n_ch, c.K = B, H, W, NO, nh, sum(nc), st['inds'].shape[2]
            c.hm_c0, c.c0_center, c.c0_z, c.c0_dim = groups['hm'][0], groups['center'][0], groups['center_z'][0], groups['dim'][0]
            c.c0_rot, c.c0_vel, c.c0_iou = groups['rot'][0], groups['vel'][0], groups['iou'][0]
            for i, h in enumerate([h for h, n in enumerate(nc) for _ in range(n)]):
                c.head_of_ch[i] = h
            for i, w in enumerate(lw['code_weights']):
                c.code_w[i] = float(w)
            c.cls_w, c.loc_w = float(lw['cls_weight']), float(lw['loc_weight'])
            c.stride, c.vs_x, c.vs_y = float(int(self.feature_map_stride)), float(self.voxel_size[0]), float(self.voxel_size[1])
            c.org_x, c.org_y = float(int(self.point_cloud_range[0])), float(int(self.point_cloud_range[1]))   # the reference's int() truncation
            cached = self._loss_cfg_cache = (key, c)
        maps = o4 if o4.is_contiguous() else o4.contiguous()
        return A.center_loss(maps, cached[1], st['heatmaps'].contiguous(), st['inds'].contiguous(), st['masks'].contiguous(),
                             st['target_boxes'].contiguous(), st['gt_box'].contiguous())

    def get_loss_per_head(self):
        pred_dicts = self.forward_ret_dict['pred_dicts']
        target_dicts = self.forward_ret_dict['target_dicts']
        tb_dict = {}
        loss = 0
        lw = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        for idx, pred_dict in enumerate(pred_dicts):
            hm = self.sigmoid(pred_dict['hm'])
            hm_loss = self.hm_loss_func(hm, target_dicts['heatmaps'][idx]) * lw['cls_weight']
            target_boxes = target_dicts['target_boxes'][idx]
            pred_boxes = torch.cat([pred_dict[head_name] for head_name in self.separate_head_cfg.HEAD_ORDER], dim=1)
            if self.with_iou and self.with_iou_reg:
                pred_boxes = pred_boxes[:, :-1, :, :]
            reg_loss = self.reg_loss_func(pred_boxes, target_dicts['masks'][idx], target_dicts['inds'][idx], target_boxes)
            loc_loss = (reg_loss * reg_loss.new_tensor(lw['code_weights'])).sum() * lw['loc_weight']
            loss = loss + hm_loss + loc_loss
            tb_dict['hm_loss_head_%d' % idx] = hm_loss.detach()
            tb_dict['loc_loss_head_%d' % idx] = loc_loss.detach()
            if self.with_iou or self.with_iou_reg:
                batch_dim = torch.exp(torch.clamp(pred_dict['dim'], min=-5, max=5))
                batch_rot = torch.atan2(pred_dict['rot'][:, 1:2], pred_dict['rot'][:, 0:1])
                B, _, H, W = batch_dim.shape
                ys, xs = torch.meshgrid(torch.arange(0, H, device=batch_dim.device), torch.arange(0, W, device=batch_dim.device), indexing='ij')
                xs = xs.view(1, 1, H, W).to(batch_dim) + pred_dict['center'][:, 0:1]
                ys = ys.view(1, 1, H, W).to(batch_dim) + pred_dict['center'][:, 1:2]
                # parity trap kept: int() truncates the range origin (radar_center_head.py:309-310)
                xs = xs * int(self.feature_map_stride) * self.voxel_size[0] + int(self.point_cloud_range[0])
                ys = ys * int(self.feature_map_stride) * self.voxel_size[1] + int(self.point_cloud_range[1])
                batch_box_preds = torch.cat([xs, ys, pred_dict['center_z'], batch_dim, batch_rot], dim=1)
                if self.with_iou:
                    iou_loss = self.crit_iou(pred_dict['iou'], target_dicts['masks'][idx], target_dicts['inds'][idx],
                                             batch_box_preds.detach(), target_dicts['gt_box'][idx])
                    loss = loss + iou_loss
                    tb_dict['iou_loss_head_%d' % idx] = iou_loss.detach()
                if self.with_iou_reg:
                    iou_reg_loss = self.crit_iou_reg(batch_box_preds, target_dicts['masks'][idx], target_dicts['inds'][idx],
                                                     target_dicts['gt_box'][idx])
                    loss = loss + lw['loc_weight'] * iou_reg_loss
                    tb_dict['iou_reg_loss_head_%d' % idx] = iou_reg_loss.detach()
        tb_dict['rpn_loss'] = loss.detach()
        return loss, tb_dict

    def generate_predicted_boxes(self, batch_size, pred_dicts):
        """Eval-time decode + rotated NMS (radar_center_head.py:332-389): top-K peaks per head, box decode, score / range filter,
        class-agnostic NMS on the device (rd_nms_bev), labels mapped to global ids (+1)."""
        from ..model_utils import model_nms_utils
        post_process_cfg = self.model_cfg.POST_PROCESSING
        dev = pred_dicts[0]['hm'].device
        post_center_limit_range = torch.tensor(post_process_cfg.POST_CENTER_LIMIT_RANGE, device=dev).float()
        ret_dict = [{'pred_boxes': [], 'pred_scores': [], 'pred_labels': []} for _ in range(batch_size)]
        for idx, pred_dict in enumerate(pred_dicts):
            batch_hm = pred_dict['hm'].sigmoid()
            batch_dim = pred_dict['dim'].exp()
            batch_iou = None
            if 'iou' in pred_dict.keys():
                batch_iou = ((pred_dict['iou'].squeeze(dim=-1) + 1) * 0.5).type_as(batch_dim)
            batch_rot_cos = pred_dict['rot'][:, 0].unsqueeze(dim=1)
            batch_rot_sin = pred_dict['rot'][:, 1].unsqueeze(dim=1)
            batch_vel = pred_dict['vel'] if 'vel' in self.separate_head_cfg.HEAD_ORDER else None
            final_pred_dicts = centernet_utils.decode_bbox_from_heatmap(
                heatmap=batch_hm, rot_cos=batch_rot_cos, rot_sin=batch_rot_sin, center=pred_dict['center'],
                center_z=pred_dict['center_z'], dim=batch_dim, vel=batch_vel, iou=batch_iou,
                rectifier=self.model_cfg.get("RECTIFIER", 0.), point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size,
                feature_map_stride=self.feature_map_stride, K=post_process_cfg.MAX_OBJ_PER_SAMPLE,
                circle_nms=(post_process_cfg.NMS_CONFIG.NMS_TYPE == 'circle_nms'), score_thresh=post_process_cfg.SCORE_THRESH,
                post_center_limit_range=post_center_limit_range)
            id_map = self.class_id_mapping_each_head[idx].to(dev)
            for k, final_dict in enumerate(final_pred_dicts):
                final_dict['pred_labels'] = id_map[final_dict['pred_labels'].long()]
                if post_process_cfg.NMS_CONFIG.NMS_TYPE != 'circle_nms':
                    selected, selected_scores = model_nms_utils.class_agnostic_nms(
                        box_scores=final_dict['pred_scores'], box_preds=final_dict['pred_boxes'],
                        nms_config=post_process_cfg.NMS_CONFIG, score_thresh=None)
                    final_dict['pred_boxes'] = final_dict['pred_boxes'][selected]
                    final_dict['pred_scores'] = selected_scores
                    final_dict['pred_labels'] = final_dict['pred_labels'][selected]
                ret_dict[k]['pred_boxes'].append(final_dict['pred_boxes'])
                ret_dict[k]['pred_scores'].append(final_dict['pred_scores'])
                ret_dict[k]['pred_labels'].append(final_dict['pred_labels'])
        for k in range(batch_size):
            ret_dict[k]['pred_boxes'] = torch.cat(ret_dict[k]['pred_boxes'], dim=0)
            ret_dict[k]['pred_scores'] = torch.cat(ret_dict[k]['pred_scores'], dim=0)
            ret_dict[k]['pred_labels'] = torch.cat(ret_dict[k]['pred_labels'], dim=0) + 1
        return ret_dict

    # ------------------------------------------------------------------ forward
    def head_forward(self, spatial_features_2d):
        x = D.conv_bn_act(spatial_features_2d, self.shared_conv[0], self.shared_conv[1], None, act=1, return_rows=True)
        self.forward_ret_dict.pop('pred_stacked', None)
        if x[0].is_cuda and self.model_cfg.get('BATCH_BRANCHES', True):
            if getattr(self, '_branch_plan', None) is None:
                self._branch_plan = (_BranchBatch.build(self),)          # (None,) = not batchable, decided once
            plan = self._branch_plan[0]
            res = plan.run(x) if plan is not None else None
            if res is not None:
                self.forward_ret_dict['pred_stacked'] = (res[0], plan.groups)
                return res[1]
        return [head(x) for head in self.heads_list]

    def forward(self, data_dict):
        spatial_features_2d = data_dict[self.FEATURE_KEY]
        pred_dicts = self.head_forward(spatial_features_2d)
        if self.training:
            if data_dict['gt_boxes'].is_cuda and self.model_cfg.TARGET_ASSIGNER_CONFIG.get('DEVICE', 'gpu') == 'gpu' and not self.waymo:
                target_dict = self.assign_targets_gpu(data_dict['gt_boxes'], spatial_features_2d.size()[2:])
            else:
                target_dict = self.assign_targets(data_dict['gt_boxes'], feature_map_size=spatial_features_2d.size()[2:],
                                                  gt_boxes_host=data_dict.get('gt_boxes_host', None))
            self.forward_ret_dict['target_dicts'] = target_dict
            if self.model_cfg.get('DISTILL_PRED', None) and not self.IS_TEACHER:
                data_dict['target_dicts'] = target_dict
                data_dict['radar_pred_dicts'] = pred_dicts
        if self.IS_TEACHER and self.model_cfg.get('DISTILL_PRED', None):       # center_head.py:403-406
            data_dict['lidar_pred_dicts'] = pred_dicts
            return data_dict
        self.forward_ret_dict['pred_dicts'] = pred_dicts
        if self.predict_boxes_when_training and self.training:
            raise NotImplementedError("ROI refinement (two-stage heads) is out of scope: SURVEY section 2")
        if not self.training:
            data_dict['final_box_dicts'] = self.generate_predicted_boxes(data_dict['batch_size'], pred_dicts)
        return data_dict
