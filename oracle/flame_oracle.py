"""CPU restatement of the FLAME vertex path (reference app/flame_model/FLAME.py:117-142 -> app/flame_model/lbs.py).
TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/flame_lbs.npz, which oracle/make_golden_flame.py produces by calling the
reference's own ``lbs()`` (imported from /root/reference/app/flame_model/lbs.py) on a deterministic synthetic asset: the real
``FLAME_with_eye.pt`` is licence-gated and absent."""
import torch


def batch_rodrigues(rot_vecs):
    # lbs.py:266-296
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos, sin = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros(n, 1)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    return torch.eye(3)[None] + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    # lbs.py:313-373
    joints = joints[..., None]
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    T = torch.cat([torch.nn.functional.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                   torch.nn.functional.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1)], dim=2).reshape(-1, joints.shape[1], 4, 4)
    chain = [T[:, 0]]
    for i in range(1, parents.shape[0]):
        chain.append(torch.matmul(chain[parents[i]], T[:, i]))
    G = torch.stack(chain, dim=1)
    jh = torch.nn.functional.pad(joints, [0, 0, 0, 1])
    return G[:, :, :3, 3], G - torch.nn.functional.pad(torch.matmul(G, jh), [3, 0, 0, 0, 0, 0, 0, 0])


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    # lbs.py:142-233 (pose2rot=True)
    B = betas.shape[0]
    v_shaped = v_template[None] + torch.einsum("bl,mkl->bmk", betas, shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    rot = batch_rodrigues(pose.view(-1, 3)).view(B, -1, 3, 3)
    feat = (rot[:, 1:] - torch.eye(3)).view(B, -1)
    v_posed = torch.matmul(feat, posedirs).view(B, -1, 3) + v_shaped
    _, A = batch_rigid_transform(rot, J, parents)
    T = torch.matmul(lbs_weights[None].expand(B, -1, -1), A.view(B, J_regressor.shape[0], 16)).view(B, -1, 4, 4)
    homo = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1)], dim=2)
    return torch.matmul(T, homo[..., None])[:, :, :3, 0]


def flame_forward(asset, shape_params, expression_params, pose_params, n_shape=300, n_exp=100, scale=1.0):
    # FLAME.py:33-45 (buffers) and :117-142 (forward, no_lmks=True)
    fm = asset["flame_model"]
    sd = fm["shapedirs"].float()
    shapedirs = torch.cat([sd[:, :, :n_shape], sd[:, :, 300:300 + n_exp]], 2)
    pd = fm["posedirs"].float()
    posedirs = pd.reshape(-1, pd.shape[-1]).T
    parents = fm["kintree_table"][0].clone()
    parents[0] = -1
    n = shape_params.shape[0]
    betas = torch.cat([shape_params, expression_params], dim=1)
    full_pose = torch.cat([pose_params[:, :3], torch.zeros(n, 3), pose_params[:, 3:], torch.zeros(n, 6)], dim=1)
    return lbs(betas, full_pose, fm["v_template"].float(), shapedirs, posedirs, fm["J_regressor"].float(), parents, fm["weights"].float()) * scale
