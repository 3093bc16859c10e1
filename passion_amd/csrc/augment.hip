// GPU input pipeline of the training loader: what the reference does per sample on the host with numpy /
// scipy -- code/data/transforms.py:86-120 (RandomRotion: scipy.ndimage.rotate per channel, order 0, reshape=False,
// mode='constant', cval=-1), :133-155 (RandomFlip), :217-250 (RandCrop3D, RandomIntensityChange) and
// code/data/datasets_nii.py:143-153 (HWDC -> CHWD, one-hot float64) -- as ONE gather kernel over the cropped volume:
// every output voxel walks the chain backwards (flip -> intensity row -> rotation -> crop voxel).
// Nearest-neighbour rotation follows scipy's affine_transform exactly: input coordinate cc = offset + o0*m0 + o1*m1 in
// double (left-to-right accumulation), outside if cc < 0 or cc > n-1, else index floor(cc + 0.5).  Rotated-out voxels
// take cval = -1; for the LABEL volume scipy casts that -1 to the array's dtype: 0 (background) for the uint8 volumes
// preprocess_brats.py:78 writes, -1 for signed types -- which np.eye(4)[y] then reads as class 3 (seg_cval carries it).
#include "common.h"
#include "passion_hip.h"

namespace {

struct AugP {
    int S0, S1, S2, C, ncls;
    int a0, a1, do_rot;        // rotation plane (sorted axes of [H, W, D]); do_rot = 0: identity
    double m00, m01, m10, m11, off0, off1;
    int f0, f1, f2;            // flips
    int seg_cval;              // label of a rotated-out voxel (see above)
};

__global__ __launch_bounds__(256) void augment_kernel(AugP p, const float* __restrict__ vol, const int* __restrict__ seg,
                                                      const double* __restrict__ scale, const double* __restrict__ shift,
                                                      float* __restrict__ x, double* __restrict__ y, float* __restrict__ ycl) {
    const long V = (long)p.S0 * p.S1 * p.S2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
        int o[3];
        o[2] = (int)(i % p.S2);
        o[1] = (int)((i / p.S2) % p.S1);
        o[0] = (int)(i / ((long)p.S2 * p.S1));
        // undo RandomFlip: flipped[i] = img[n-1-i]
        int r[3] = {p.f0 ? p.S0 - 1 - o[0] : o[0], p.f1 ? p.S1 - 1 - o[1] : o[1], p.f2 ? p.S2 - 1 - o[2] : o[2]};
        const int hrow = r[0];                       // RandomIntensityChange draws one (shift, scale) per H index and channel
        // undo RandomRotion: rotated[r] = crop[nearest(M r + off)] or cval
        int s[3] = {r[0], r[1], r[2]};
        bool inside = true;
        if (p.do_rot) {
            const int n[3] = {p.S0, p.S1, p.S2};
            const double o0 = (double)r[p.a0], o1 = (double)r[p.a1];
            double c0 = p.off0 + o0 * p.m00; c0 = c0 + o1 * p.m01;
            double c1 = p.off1 + o0 * p.m10; c1 = c1 + o1 * p.m11;
            inside = !(c0 < 0.0 || c0 > (double)(n[p.a0] - 1) || c1 < 0.0 || c1 > (double)(n[p.a1] - 1));
            if (inside) { s[p.a0] = (int)floor(c0 + 0.5); s[p.a1] = (int)floor(c1 + 0.5); }
        }
        const long sv = ((long)s[0] * p.S1 + s[1]) * p.S2 + s[2];
        for (int c = 0; c < p.C; ++c) {
            const double v = inside ? (double)vol[sv * p.C + c] : -1.0;
            const double t = v * scale[hrow * p.C + c] + shift[hrow * p.C + c];     // float32 * float64 + float64, as numpy
            x[(long)c * V + i] = (float)t;
        }
        int lab = inside ? seg[sv] : p.seg_cval;
        if (lab < 0) lab += p.ncls;                  // numpy negative index
        for (int c = 0; c < p.ncls; ++c) {
            const float hot = (c == lab) ? 1.f : 0.f;
            if (y) y[(long)c * V + i] = (double)hot;
            if (ycl) ycl[i * p.ncls + c] = hot;
        }
    }
}

}  // namespace

extern "C" int passion_augment(const float* crop, const int32_t* seg, int S0, int S1, int S2, int C, int num_cls, int rot_a0, int rot_a1,
                               const double* rot6, int seg_cval, int flip0, int flip1, int flip2, const double* scale, const double* shift,
                               float* x, double* y_onehot, float* y_cl, void* stream) {
    if (S0 < 1 || S1 < 1 || S2 < 1 || C < 1 || num_cls < 1) return -1041;
    AugP p;
    p.S0 = S0; p.S1 = S1; p.S2 = S2; p.C = C; p.ncls = num_cls;
    p.do_rot = rot6 != nullptr;
    p.a0 = rot_a0; p.a1 = rot_a1;
    if (p.do_rot) {
        if (!(rot_a0 >= 0 && rot_a0 < rot_a1 && rot_a1 <= 2)) return -1042;
        p.m00 = rot6[0]; p.m01 = rot6[1]; p.m10 = rot6[2]; p.m11 = rot6[3]; p.off0 = rot6[4]; p.off1 = rot6[5];
    } else {
        p.m00 = p.m11 = 1.0; p.m01 = p.m10 = p.off0 = p.off1 = 0.0;
    }
    p.f0 = flip0; p.f1 = flip1; p.f2 = flip2;
    p.seg_cval = seg_cval;
    const long V = (long)S0 * S1 * S2;
    const int blocks = (int)hmin((V + 255) / 256, 8192);
    hipLaunchKernelGGL(augment_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, crop, seg, scale, shift, x, y_onehot, y_cl);
    LAUNCH_RET();
    return 0;
}
