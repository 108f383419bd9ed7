"""The composed hot path: PointNet++ point encoder -> situational pose re-encode -> Q-Former
fusion, behind the reference's `forward(data_dict) -> data_dict` model API
(situation3d/models/sqa_module.py:281-392; keys listed in SURVEY.md section 3.1).

The three stages exist in the reference in three disjoint places, so the
COMPOSITION is this build's; each stage is parity-checked against the reference code of that
stage.  Shapes follow the north star: N = 40 000 points / scene -> SA1..SA4 (VoteNet lineage:
2048/0.2/64, 1024/0.4/32, 512/0.8/16, 256/1.2/16) -> 256 visual tokens x 256 channels, which is
exactly SIG3D's visual-token interface (lib/config.py:104-105); 32 learned queries + the
question tokens attend to them in a BLIP-2 Q-Former (cross-attention every 2nd layer).
"""
import torch
import torch.nn as nn

from .pointnet2.fused_mlp import attach_scan, point_major_of
from .pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes
from .qformer import init_Qformer
from .situational import gaussian_localisation_target, situational_transform
from .small_mlp import pos_embed_add, posed_pos_embed_add
from . import heads


class PointNet2Encoder(nn.Module):
    """4 x SA (+ optional 2 x FP) backbone; SA hyper-parameters as in SURVEY.md section 8."""

    def __init__(self, input_feature_dim=3, use_fp=False):
        super().__init__()
        self.sa1 = PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64,
                                         mlp=[input_feature_dim, 64, 64, 128], use_xyz=True,
                                         normalize_xyz=True)
        self.sa2 = PointnetSAModuleVotes(npoint=1024, radius=0.4, nsample=32,
                                         mlp=[128, 128, 128, 256], use_xyz=True, normalize_xyz=True)
        self.sa3 = PointnetSAModuleVotes(npoint=512, radius=0.8, nsample=16,
                                         mlp=[256, 128, 128, 256], use_xyz=True, normalize_xyz=True)
        self.sa4 = PointnetSAModuleVotes(npoint=256, radius=1.2, nsample=16,
                                         mlp=[256, 128, 128, 256], use_xyz=True, normalize_xyz=True)
        for sa in (self.sa1, self.sa2, self.sa3, self.sa4):
            sa.emit_point_major = True   # pooled features also point-major: no transpose between the levels
        self.use_fp = use_fp
        if use_fp:
            self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
            self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    LEVELS = [(2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)]

    def forward(self, xyz, features, plan=None):
        """xyz (B,N,3), features (B,C,N) -> token xyz (B,T,3), token features (B,256,T).
        `plan` optionally carries the precomputed geometry of the four levels (FPS indices, centre
        coordinates, ball-query lists: functions of xyz only -- geometry.GeometryPlan)."""
        g = [plan.level(i) if plan is not None else None for i in range(4)]
        xyz1, f1, i1 = self.sa1(xyz, features, geometry=g[0])
        xyz2, f2, i2 = self.sa2(xyz1, f1, geometry=g[1])
        xyz3, f3, i3 = self.sa3(xyz2, f2, geometry=g[2])
        xyz4, f4, i4 = self.sa4(xyz3, f3, geometry=g[3])
        if not self.use_fp:
            return xyz4, f4
        f3u = self.fp1(xyz3, xyz4, f3, f4)
        f2u = self.fp2(xyz2, xyz3, f2, f3u)
        return xyz2, f2u


def build_pos_embed(in_dim, feat_dim):
    """The positional MLP of sqa_module.py:274-278: Linear(in_dim,128)-GELU-Linear(128,feat_dim).  The reference
    feeds it the 2-D (x, y) token positions in the WORLD frame (:319-321, in_dim = 2); the north-star path feeds
    it the 3-D token positions in the AGENT's frame (situational re-encode, in_dim = 3, the default here)."""
    return nn.Sequential(nn.Linear(in_dim, 128), nn.GELU(), nn.Linear(128, feat_dim))


class SIG3DQFormer(nn.Module):
    """data_dict in : point_clouds (B,N,3+C) f32 [xyz | per-point features],
                      auxiliary_task (B,7) f32 [x,y,z, quat_xyzw] (situation; sepdataset.py:306-315),
                      q_feat {"input_ids" (B,T) i64, "attention_mask" (B,T)} (question tokens)
       data_dict out: answer_scores (B,num_answers), aux_scores (B,7), auxiliary_task_loc_gt (B,T),
                      pred_pos_likelihood (B,T), pred_rotation (B,T,6), scene_positions (B,T,3),
                      att_feat_pre (B,T,256), att_feat_ori (B,32,768)
    """

    def __init__(self, num_answers=706, input_feature_dim=3, num_query_token=32, use_fp=False,
                 qformer_overrides=None, vocab_size=30522, pos_embed_dim=3):
        super().__init__()
        self.encoder = PointNet2Encoder(input_feature_dim, use_fp)
        feat_dim = 256
        # pos_embed_dim = 3 (default): the MLP sees the situational (agent-frame) xyz of a token;
        # pos_embed_dim = 2: the reference's layer (sqa_module.py:274-278, same state_dict keys and shapes, a
        # reference checkpoint's `pos_embed.*` loads) on the world-frame (x, y) of a token, as :319-321
        assert pos_embed_dim in (2, 3)
        self.pos_embed_dim = pos_embed_dim
        self.pos_embed = build_pos_embed(pos_embed_dim, feat_dim)
        overrides = dict(vocab_size=vocab_size)
        overrides.update(qformer_overrides or {})
        self.Qformer, self.query_tokens = init_Qformer(num_query_token, feat_dim, **overrides)
        hidden = self.Qformer.config.hidden_size
        self.position_head = nn.Linear(feat_dim, 1)
        self.rotation_head = nn.Linear(feat_dim, 6)
        self.aux_reg = nn.Sequential(nn.Linear(hidden, hidden), nn.GELU(), nn.Linear(hidden, 7))
        self.answer_cls = nn.Sequential(nn.Linear(hidden, hidden), nn.GELU(), nn.Dropout(0.1),
                                        nn.Linear(hidden, num_answers))

    def forward(self, data_dict):
        pc = data_dict["point_clouds"]
        xyz = pc[..., :3].contiguous()
        # a channel-major VIEW of the colours (no copy): SA1's first layer reads the raw point-major rows of the scan
        # (fused_mlp.first_layer_scan); any other path makes its own contiguous copy
        features = None
        if pc.shape[-1] > 3:
            features = attach_scan(pc[..., 3:].transpose(1, 2), pc) if pc.is_cuda else pc[..., 3:].transpose(1, 2).contiguous()
        tok_xyz, tok_feat = self.encoder(xyz, features, data_dict.get("geometry_plan"))
        tok_pm = point_major_of(tok_feat)                           # SA4's pooling kernel wrote (B,T,256) as well
        tok_feat = tok_pm if tok_pm is not None else tok_feat.transpose(1, 2).contiguous()
        data_dict["scene_positions"] = tok_xyz
        data_dict["att_feat_pre"] = tok_feat

        # situational re-encode: token positions in the agent's frame, R(q)^T (p - t)
        pose = data_dict["auxiliary_task"]
        data_dict["auxiliary_task_loc_gt"] = gaussian_localisation_target(tok_xyz, pose)   # reads pose[:, :2]
        # re-encode + tok_feat + Linear(GELU(Linear(position))) as ONE launch (small_mlp.posed_pos_embed_add: the
        # transform is formed inside the positional MLP's kernel); two launches / the torch path where that is not covered
        fused = posed_pos_embed_add(self.pos_embed, pose, tok_xyz, tok_feat, inverse=True) if self.pos_embed_dim == 3 else None
        if fused is not None:
            tokens, sit_xyz = fused
        else:
            sit_xyz = situational_transform(pose, tok_xyz, inverse=True)
            tokens = pos_embed_add(self.pos_embed, sit_xyz if self.pos_embed_dim == 3 else tok_xyz[..., :2], tok_feat)
        data_dict["situational_positions"] = sit_xyz
        if data_dict.get("_split_backward"):
            # data-parallel step (graph_step.py): the backward pass is cut at the visual tokens (and, with
            # `_qf_cut`, once more inside the Q-Former) so that the gradient all-reduce of everything
            # downstream of a cut overlaps the backward of everything upstream of it
            leaf = tokens.detach().requires_grad_(True)
            data_dict["_boundary"] = (tokens, leaf)
            tokens = leaf
            self.Qformer.bert.encoder.cut_after = data_dict.get("_qf_cut")

        data_dict["pred_pos_likelihood"] = self.position_head(tokens).squeeze(-1)
        data_dict["pred_rotation"] = self.rotation_head(tokens)

        q = data_dict.get("q_feat")
        b = tokens.shape[0]
        query_tokens = self.query_tokens.expand(b, -1, -1)
        kwargs = {}
        if q is not None:
            ones = torch.ones(b, query_tokens.shape[1], dtype=q["attention_mask"].dtype,
                              device=tokens.device)
            kwargs = dict(input_ids=q["input_ids"],
                          attention_mask=torch.cat([ones, q["attention_mask"]], dim=1))
        out = self.Qformer.bert(query_embeds=query_tokens, encoder_hidden_states=tokens,
                                encoder_attention_mask=None, return_dict=True, **kwargs)
        if data_dict.get("_split_backward"):
            data_dict["_qf_boundary"] = self.Qformer.bert.encoder.cut   # None when no cut was requested / taken
            self.Qformer.bert.encoder.cut = None
        fused = getattr(out, "query_hidden_state", None)   # two-segment layout: a free view
        if fused is None:
            fused = out.last_hidden_state[:, :query_tokens.shape[1], :]
        data_dict["att_feat_ori"] = fused
        rows = getattr(out, "rows", None)      # the two-segment row matrix `fused` is a view of (qformer._SegmentedOutput)
        if rows is None and fused.is_contiguous():
            rows = fused.view(-1, fused.shape[-1])
        if rows is not None and heads.covered(rows, b, fused.shape[1], self.aux_reg, self.answer_cls):
            # mean over the query rows + both heads: three launches, two in the backward pass (csrc/heads.hip)
            data_dict["aux_scores"], data_dict["answer_scores"] = heads.pooled_heads(rows, b, fused.shape[1], self.aux_reg,
                                                                                    self.answer_cls)
        else:
            pooled = fused.mean(dim=1)
            data_dict["aux_scores"] = self.aux_reg(pooled)
            data_dict["answer_scores"] = self.answer_cls(pooled)
        return data_dict
