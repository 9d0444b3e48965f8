"""A second, differently built restatement of the shader, written from shaders/compute.comp and shaders/include/*.h alone (not from
oracle/oracle.c): float64 arithmetic with libm's pow / sin / cos, a brute-force closest hit over all triangles instead of any
tree, Vulkan's unnormalised-coordinate bilinear fetch.  It shares no code, no polynomial and no data structure with the oracle; what
it shares is the reading of the GLSL's *composition*: which random number is drawn where, which `seed` arguments are copies
(generate_camera_ray and sample_GGXVNDF take `uint seed` by value, rand.h's functions take it `inout`), which branch ends a path
with black and which with the throughput, how a frame is blended into the accumulation image.  The oracle's frame must agree with
it to float32 accuracy on all but a handful of pixels (a path whose discrete decision -- a hit at a triangle edge, `rand < p` --
falls the other way at the other precision)."""
import math

import numpy as np

from common import make_bundle, oracle_frames

U32 = 0xFFFFFFFF
TWO_PI = 2.0 * 3.141592653589793


# ---- rand.h:1-12 (seed `inout`: the caller's state advances) ----
class Seed:
    def __init__(self, state):
        self.s = state & U32

    def copy(self):
        return Seed(self.s)


def pcg_hash(seed):
    state = seed.s
    seed.s = (seed.s * ((747796405 + 2891336453) & U32)) & U32       # `seed *= a + b` multiplies by the SUM
    word = ((((state >> ((state >> 28) + 4)) ^ state) & U32) * 277803737) & U32
    return ((word >> 22) ^ word) & U32


def rand(seed):
    return float(np.float32(pcg_hash(seed))) / 4294967296.0          # float(uint) rounds to fp32, the division is exact


# ---- color_utils.h ----
def luminance(c):
    return 0.2126 * c[0] + 0.7152 * c[1] + 0.0722 * c[2]


def srgb_to_linear(c):
    c = np.clip(c, 0.0, 1.0)
    return np.where(c < 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def linear_to_srgb(c):
    c = np.clip(c, 0.0, 1.0)
    return np.where(c < 0.0031308, c * 12.92, c ** (1.0 / 2.4) * 1.055 - 0.055)


# ---- math.h ----
def normalize(v):
    return v / math.sqrt(float(np.dot(v, v)))


def rotation_to_z_axis(axis):
    if axis[2] < -0.99999:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return normalize(np.array([axis[1], -axis[0], 0.0, 1.0 + axis[2]]))


def rotate_point(q, v):
    qa = q[:3]
    return 2.0 * np.dot(qa, v) * qa + (q[3] * q[3] - np.dot(qa, qa)) * v + 2.0 * q[3] * np.cross(qa, v)


def sample_hemisphere(r0, r1):
    radius, theta = math.sqrt(r0), TWO_PI * r1
    return np.array([math.cos(theta) * radius, math.sin(theta) * radius, math.sqrt(1.0 - r0)])


def disk_vec(r0, r1):
    z, a = r0 * 2.0 - 1.0, TWO_PI * r1
    r = math.sqrt(1.0 - z * z)
    return np.array([math.cos(a) * r, math.sin(a) * r])


# ---- brdf.h ----
def sample_ggx_vndf(ve, ax, ay, seed_by_value):
    seed = seed_by_value.copy()                                        # `uint seed`: the caller's state does NOT advance
    vh = normalize(np.array([ax * ve[0], ay * ve[1], ve[2]]))
    lensq = vh[0] * vh[0] + vh[1] * vh[1]
    t1v = np.array([-vh[1], vh[0], 0.0]) / math.sqrt(lensq) if lensq > 0.0 else np.array([1.0, 0.0, 0.0])
    t2v = np.cross(vh, t1v)
    r = math.sqrt(rand(seed))
    phi = TWO_PI * rand(seed)
    t1, t2 = r * math.cos(phi), r * math.sin(phi)
    s = 0.5 * (1.0 + vh[2])
    t2 = math.sqrt(1.0 - t1 * t1) * (1.0 - s) + t2 * s
    nh = t1 * t1v + t2 * t2v + math.sqrt(max(0.0, 1.0 - t1 * t1 - t2 * t2)) * vh
    return normalize(np.array([ax * nh[0], ay * nh[1], max(0.0, nh[2])]))


def fresnel_schlick(f0, f90, n_dot_s):
    return f0 + (f90 - f0) * (1.0 - n_dot_s) ** 5.0


def shadowed_f90(f0):
    return min(1.0, (1.0 / 0.04) * luminance(f0))


def brdf_probability(color, metalness, view, normal):
    f0 = luminance(0.04 * (1.0 - metalness) + color * metalness)
    specular_f0 = np.array([f0, f0, f0])
    diffuse_reflectance = luminance(color * (1.0 - metalness))
    fresnel = min(1.0, max(0.0, luminance(fresnel_schlick(specular_f0, shadowed_f90(specular_f0), max(0.0, float(np.dot(view, normal)))))))
    specular, diffuse = fresnel, diffuse_reflectance * (1.0 - fresnel)
    return min(0.9, max(0.1, specular / max(0.0001, specular + diffuse)))


# ---- texture(sampler2D, uv): LINEAR, REPEAT, lod 0, RGBA8 UNORM, unnormalised coordinate u * size - 0.5 ----
def texture(tex, u, v):
    if tex is None:
        return np.ones(4)                                              # this repository's definition of texture id 0
    h, w = tex.shape[:2]
    x, y = (u - math.floor(u)) * w - 0.5, (v - math.floor(v)) * h - 0.5
    i0, j0 = math.floor(x), math.floor(y)
    a, b = x - i0, y - j0
    t = tex.astype(np.float64) / 255.0
    c = lambda i, j: t[j % h, i % w]
    return (c(i0, j0) * (1 - a) + c(i0 + 1, j0) * a) * (1 - b) + (c(i0, j0 + 1) * (1 - a) + c(i0 + 1, j0 + 1) * a) * b


class Scene:
    def __init__(self, b, materials, textures):
        idx = np.asarray(b["indices"], np.uint32).reshape(-1, 3)
        self.vid = (idx & 0x00FFFFFF).astype(np.int64)
        self.mat_id = (((idx[:, 0] & 0xFF000000) >> 8) | ((idx[:, 1] & 0xFF000000) >> 16) | ((idx[:, 2] & 0xFF000000) >> 24)).astype(np.int64)
        self.pos = np.asarray(b["positions"], np.float64).reshape(-1, 3)
        self.nrm = np.asarray(b["normals"], np.float64).reshape(-1, 3)
        self.uv = np.asarray(b["uvs"], np.float64).reshape(-1, 2)
        self.p0 = self.pos[self.vid[:, 0]]
        self.e1 = self.pos[self.vid[:, 1]] - self.p0
        self.e2 = self.pos[self.vid[:, 2]] - self.p0
        self.n = np.cross(self.e1, self.e2)
        self.materials, self.textures = materials, textures

    def closest_hit(self, o, d, min_t=0.001, max_t=1e15):
        """ray.h:137-163 on every triangle at once; the smallest accepted t."""
        with np.errstate(divide="ignore", invalid="ignore"):
            ov = o - self.p0
            q = np.cross(ov, d)
            dd = 1.0 / (self.n @ d)
            u = dd * np.einsum("ij,ij->i", -q, self.e2)
            v = dd * np.einsum("ij,ij->i", q, self.e1)
            t = dd * np.einsum("ij,ij->i", -self.n, ov)
            ok = ~((u < 0.0) | (u > 1.0) | (v < 0.0) | (u + v > 1.0) | (t < min_t) | (t > max_t)) & np.isfinite(t)
        if not ok.any():
            return None
        k = int(np.argmin(np.where(ok, t, np.inf)))
        return k, float(t[k]), float(u[k]), float(v[k])


def ray_color(sc, o, d, seed, max_bounce, min_bounce):
    """compute.comp:124-245.  `radiance` is never written: only a path that escapes returns light (its throughput)."""
    throughput = np.ones(3)
    for bounce in range(max_bounce):
        hit = sc.closest_hit(o, d)
        if hit is None:
            return throughput
        k, t, u, v = hit
        view = -d
        bary = np.array([1.0 - u - v, u, v])
        tri = sc.vid[k]
        uv = sc.uv[tri[0]] * bary[0] + sc.uv[tri[1]] * bary[1] + sc.uv[tri[2]] * bary[2]
        shading_normal = normalize(sc.nrm[tri[0]] * bary[0] + sc.nrm[tri[1]] * bary[1] + sc.nrm[tri[2]] * bary[2])
        geometry_normal = normalize(sc.n[k])
        m = sc.materials[sc.mat_id[k]]
        albedo = texture(sc.textures[int(m["albedo_tex"])], uv[0], uv[1])
        diffuse_color = srgb_to_linear(albedo[:3]) * np.asarray(m["base_color"][:3], np.float64)
        mr = texture(sc.textures[int(m["mr_tex"])], uv[0], uv[1])
        metalness, roughness = mr[0] * float(m["metalness"]), mr[1] * float(m["roughness"])   # .xy: red and green
        if np.dot(geometry_normal, view) < 0.0:
            geometry_normal = -geometry_normal
        if np.dot(geometry_normal, shading_normal) < 0.0:
            shading_normal = -shading_normal
        if metalness == 1.0 and roughness == 0.0:
            specular = True
        else:
            p = brdf_probability(diffuse_color, metalness, view, shading_normal)
            if rand(seed) < p:
                specular, throughput = True, throughput / p
            else:
                specular, throughput = False, throughput / (1.0 - p)
        if np.dot(shading_normal, view) <= 0.0:
            break
        to_z = rotation_to_z_axis(shading_normal)
        view_local = rotate_point(to_z, view)
        if not specular:
            r0 = rand(seed)
            r1 = rand(seed)
            dir_local = sample_hemisphere(r0, r1)
            weight = diffuse_color * (1.0 - metalness)
        else:
            alpha = roughness * roughness
            half = np.array([0.0, 0.0, 1.0]) if alpha == 0.0 else sample_ggx_vndf(view_local, alpha, alpha, seed)
            incident = -view_local
            light = incident - 2.0 * np.dot(half, incident) * half                       # reflect(I, N)
            f0 = diffuse_color * (1.0 - metalness)                                       # compute.comp:205 takes the DIFFUSE reflectance
            h_dot_l = min(1.0, max(0.00001, float(np.dot(half, light))))
            n_dot_l = min(1.0, max(0.00001, float(light[2])))
            fr = fresnel_schlick(f0, shadowed_f90(f0), h_dot_l)
            a2, n2 = alpha * alpha, n_dot_l * n_dot_l
            weight = fr * (2.0 / (math.sqrt((a2 * (1.0 - n2) + n2) / n2) + 1.0))        # Smith G1 (no height-correlated G2)
            dir_local = light
        if luminance(weight) == 0.0:
            break
        from_z = np.array([-to_z[0], -to_z[1], -to_z[2], to_z[3]])
        ray_dir = rotate_point(from_z, dir_local)
        if np.dot(geometry_normal, ray_dir) <= 0.0:
            break
        throughput = throughput * weight
        o, d = o + d * t, ray_dir                                                        # no offset along the normal
        if bounce > min_bounce:
            probability = min(0.95, luminance(throughput))
            if probability < rand(seed):
                break
            throughput = throughput / probability
    return np.zeros(3)


def render(b, materials, textures, width, height, sample_index, max_bounce, min_bounce, acc, enable_dof=False):
    cam = np.frombuffer(np.ascontiguousarray(b["meta"]).tobytes()[:128], np.float32).astype(np.float64)
    position, horizontal, vertical, first_pixel = cam[0:3], cam[16:19], cam[20:23], cam[24:27]
    # compute.comp:44-58 declares `up` at byte 32 and `right` at byte 48 of the block the host fills with right at 32, up at 48
    # (include/camera.hpp:26-28): what the shader calls cam.up is the host's right vector and the other way round
    glsl_up, glsl_right, lens_radius = cam[8:11], cam[12:15], cam[28]
    sc = Scene(b, materials, textures)
    out = np.zeros((height, width, 4))
    for y in range(height):
        for x in range(width):
            seed = Seed(((x * 1973 + y * 9277 + sample_index * 26699) & U32) | 1)
            cam_seed = seed.copy()                                                        # generate_camera_ray(coords, seed): by value
            size = np.array([width - 1.0, height - 1.0])
            uv = np.array([x / size[0], 1.0 - y / size[1]])
            r0 = rand(cam_seed)
            r1 = rand(cam_seed)
            lens = disk_vec(r0, r1) * (lens_radius if enable_dof else 0.0)                # drawn even with depth of field off
            jx = rand(cam_seed)
            jy = rand(cam_seed)
            juv = uv + np.array([jx, jy]) / size
            offset = lens[0] * glsl_right + lens[1] * glsl_up
            direction = first_pixel + juv[0] * horizontal + juv[1] * vertical - position - offset   # not normalised
            colour = ray_color(sc, position + offset, direction, seed, max_bounce, min_bounce)   # ... and ray_color restarts from the same state
            prev = srgb_to_linear(acc[y, x, :3])
            s = 1.0 / sample_index
            out[y, x, :3] = linear_to_srgb(prev * (1.0 - s) + colour * s)
            out[y, x, 3] = 1.0
    return out


import pytest


@pytest.mark.parametrize("kind,target,W,H,BOUNCES,min_bounce,dof", [("tiny", 0, 40, 24, 4, 1, 0), ("tiny", 0, 36, 20, 6, 0, 1),
                                                                     ("atrium", 1500, 32, 18, 3, 1, 0), ("street_exterior", 3000, 40, 25, 3, 2, 0),
                                                                     ("street_exterior", 2000, 28, 16, 4, 0, 1)])
def test_oracle_agrees_with_a_float64_brute_force_restatement_of_the_glsl(kind, target, W, H, BOUNCES, min_bounce, dof):
    b = make_bundle(kind, target, W, H, max_bounce=BOUNCES, min_bounce=min_bounce, enable_dof=dof)
    acc = np.zeros((H, W, 4))
    for s in (1, 2):                                                   # two frames: the second blends with weight 1/2
        acc = render(b, b["materials"], b["textures"], W, H, s, BOUNCES, min_bounce, acc, enable_dof=bool(dof))
    for use_flat in (False, True):
        ref, _, _ = oracle_frames(b, 2, use_flat=use_flat)
        err = np.abs(ref.astype(np.float64) - acc).max(axis=2)
        far = int((err > 2e-4).sum())
        assert far <= max(2, W * H // 200), f"{far} of {W * H} pixels differ by more than 2e-4 (use_flat={use_flat})"
        assert float(np.median(err)) < 2e-6, float(np.median(err))
        assert (ref[..., 3] == 1.0).all()
    assert acc[..., :3].std() > 0.05                                   # a picture, not a constant
