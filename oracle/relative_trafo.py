"""utils/relative_trafo.py of the reference restated in NumPy (oracle; test infrastructure only).

bone_rel_trafo_inv (:243-295) is on the inference path of PosePriorNetwork variants 'local' and
'local_w_xyz_loss' (nets/PosePriorNetwork.py:70-75); bone_rel_trafo (:184-240) is restated only so
the tests can use the round trip xyz -> (length, angle_x, angle_y) -> xyz as a property.
Batched float32 4x4 homogeneous matrices, same composition order as the reference.
"""
import numpy as np

F32 = np.float32

# utils/relative_trafo.py:148-181
kinematic_chain_dict = {0: 'root',
                        4: 'root', 3: 4, 2: 3, 1: 2,
                        8: 'root', 7: 8, 6: 7, 5: 6,
                        12: 'root', 11: 12, 10: 11, 9: 10,
                        16: 'root', 15: 16, 14: 15, 13: 14,
                        20: 'root', 19: 20, 18: 19, 17: 18}
kinematic_chain_list = [0, 4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13, 20, 19, 18, 17]


def _eye(B):
    return np.tile(np.eye(4, dtype=F32), (B, 1, 1))


def _rot_x(a):           # :48-57
    m = _eye(a.shape[0])
    m[:, 1, 1] = np.cos(a); m[:, 1, 2] = -np.sin(a)
    m[:, 2, 1] = np.sin(a); m[:, 2, 2] = np.cos(a)
    return m


def _rot_y(a):           # :60-69
    m = _eye(a.shape[0])
    m[:, 0, 0] = np.cos(a); m[:, 0, 2] = np.sin(a)
    m[:, 2, 0] = -np.sin(a); m[:, 2, 2] = np.cos(a)
    return m


def _trans(t):           # :84-93: translation along z only
    m = _eye(t.shape[0])
    m[:, 2, 3] = t
    return m


def _atan2(y, x):        # :28-45
    pi = F32(3.141592653589793)
    tan = np.arctan(y / (x + F32(1e-8))).astype(F32)
    tan_c = tan + np.where(x + F32(1e-8) < 0, pi, F32(0))
    tan02 = tan_c + np.where(tan_c < 0, F32(2) * pi, F32(0))
    return (tan02 + np.where(tan02 > pi, F32(-2) * pi, F32(0))).astype(F32)


def _forward(length, angle_x, angle_y, T):   # :108-118
    T_this = _trans(-length) @ (_rot_x(-angle_x) @ _rot_y(-angle_y))
    T = (T_this @ T).astype(F32)
    x0 = np.zeros((length.shape[0], 4, 1), F32)
    x0[:, 3, 0] = 1
    x = np.linalg.inv(T).astype(F32) @ x0        # tf.matrix_inverse
    return x, T


def _backward(delta_vec, T):                  # :121-146
    length = np.sqrt(delta_vec[:, 0, 0] ** 2 + delta_vec[:, 1, 0] ** 2 + delta_vec[:, 2, 0] ** 2).astype(F32)
    angle_y = _atan2(delta_vec[:, 0, 0], delta_vec[:, 2, 0])
    tmp = _rot_y(-angle_y) @ delta_vec
    angle_x = _atan2(-tmp[:, 1, 0], tmp[:, 2, 0])
    T_this = _trans(-length) @ (_rot_x(-angle_x) @ _rot_y(-angle_y))
    return length, angle_x, angle_y, (T_this @ T).astype(F32)


def _to_hom(v):
    B = v.shape[0]
    return np.concatenate([v.reshape(B, -1, 1), np.ones((B, 1, 1), F32)], 1).astype(F32)


def bone_rel_trafo_inv(coords_rel):
    """:243-295.  coords_rel [B,21,3] = (length, angle_x, angle_y) per keypoint -> xyz [B,21,3]."""
    coords_rel = np.asarray(coords_rel, dtype=F32)
    if coords_rel.ndim == 2:
        coords_rel = coords_rel[None]
    assert coords_rel.ndim == 3, "Has to be a batch of coords."
    B = coords_rel.shape[0]
    trafo = [None] * 21
    out = np.zeros((B, 21, 3), F32)
    for bone_id in kinematic_chain_list:
        parent = kinematic_chain_dict[bone_id]
        T = _trans(np.zeros(B, F32)) if parent == 'root' else trafo[parent]
        assert T is not None, 'Something went wrong.'
        x, T = _forward(coords_rel[:, bone_id, 0], coords_rel[:, bone_id, 1], coords_rel[:, bone_id, 2], T)
        out[:, bone_id, :] = x[:, :3, 0]
        trafo[bone_id] = T
    return out


def bone_rel_trafo(coords_xyz):
    """:184-240 (training-label side; here only for the round-trip property test)."""
    coords_xyz = np.asarray(coords_xyz, dtype=F32)
    if coords_xyz.ndim == 2:
        coords_xyz = coords_xyz[None]
    B = coords_xyz.shape[0]
    trafo = [None] * 21
    out = np.zeros((B, 21, 3), F32)
    for bone_id in kinematic_chain_list:
        parent = kinematic_chain_dict[bone_id]
        if parent == 'root':
            delta = _to_hom(coords_xyz[:, bone_id, :])
            T = _trans(np.zeros(B, F32))
        else:
            T = trafo[parent]
            lp = T @ _to_hom(coords_xyz[:, parent, :])
            lc = T @ _to_hom(coords_xyz[:, bone_id, :])
            delta = _to_hom((lc - lp)[:, :3, 0])
        l, ax, ay, Tn = _backward(delta, T)
        out[:, bone_id, 0], out[:, bone_id, 1], out[:, bone_id, 2] = l, ax, ay
        trafo[bone_id] = Tn
    return out
