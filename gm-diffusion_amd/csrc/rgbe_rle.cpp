// Host side of the Radiance .hdr writer: the run-length framing of RGBE scanlines.  The reference writes its HDR pictures with
// cv2.imwrite("*.hdr") (scripts/inference/generate_hdr.py:27-30); OpenCV's HDR encoder is the rgbe.c writer of the Radiance
// format and compresses scanlines by default.  cv2 is absent in this image, so this follows the published format ("new"
// adaptive run-length scheme of Radiance pictures, as written by rgbe.c's RGBE_WritePixels_RLE / RGBE_WriteBytes_RLE):
//   scanline = {2, 2, width >> 8, width & 255}, then the four components (R, G, B, E planes of the line) one after the other,
//   each as a sequence of   [128 + n, value]  (a run of n equal bytes, 4 <= n <= 127, or a short run 2..3 that fills the
//   whole gap in front of the next long run)  and  [n, b_1 .. b_n]  (n <= 128 literal bytes);
//   widths below 8 or above 32767 are written flat (4 bytes per pixel), as the format prescribes.
// Pixels come from gmd_rgbe_encode (device); this is byte shuffling on the host next to the file write, not a kernel.
#include <stdint.h>
#include <string.h>
#include "../../include/gmd_hip.h"

void gmd_set_error(const char* fmt, ...);

namespace {

constexpr int kMinRun = 4;

// one component plane of one scanline -> out; returns the bytes written
int64_t rle_bytes(const uint8_t* data, int n, uint8_t* out) {
    int64_t o = 0;
    int cur = 0;
    while (cur < n) {
        int beg_run = cur, run_count = 0, old_run_count = 0;
        while (run_count < kMinRun && beg_run < n) {  // the next run of at least kMinRun equal bytes, if there is one
            beg_run += run_count;
            old_run_count = run_count;
            run_count = 1;
            while (beg_run + run_count < n && run_count < 127 && data[beg_run] == data[beg_run + run_count]) ++run_count;
        }
        if (old_run_count > 1 && old_run_count == beg_run - cur) {  // the gap in front of it is itself one short run
            out[o++] = (uint8_t)(128 + old_run_count);
            out[o++] = data[cur];
            cur = beg_run;
        }
        while (cur < beg_run) {  // literal bytes up to the run
            int m = beg_run - cur;
            if (m > 128) m = 128;
            out[o++] = (uint8_t)m;
            memcpy(out + o, data + cur, (size_t)m);
            o += m;
            cur += m;
        }
        if (run_count >= kMinRun) {
            out[o++] = (uint8_t)(128 + run_count);
            out[o++] = data[beg_run];
            cur += run_count;
        }
    }
    return o;
}

}  // namespace

extern "C" {

int64_t gmd_rgbe_rle_bound(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    // per component at most one count byte per 128 literals (+1), runs never expand; 4 header bytes per scanline; + W bytes
    // behind the worst-case output for the encoder's one-plane scratch line (it must never overlap bytes already written:
    // for a wide, short, incompressible picture -- 1 x 1000 -- the slack of the first term alone is smaller than W)
    return (int64_t)H * (4 + 4 * ((int64_t)W + W / 128 + 2)) + W;
}

int gmd_rgbe_rle_encode(const uint8_t* rgbe, int H, int W, uint8_t* out, int64_t capacity, int64_t* out_bytes) {
    if (!out_bytes || H < 0 || W < 0 || ((H > 0 && W > 0) && (!rgbe || !out))) {
        gmd_set_error("gmd_rgbe_rle_encode: bad argument");
        return GMD_ERR_INVALID;
    }
    *out_bytes = 0;
    if (H == 0 || W == 0) return GMD_OK;
    if (capacity < gmd_rgbe_rle_bound(H, W)) {
        gmd_set_error("gmd_rgbe_rle_encode: output buffer of %lld bytes is smaller than gmd_rgbe_rle_bound = %lld", (long long)capacity,
                      (long long)gmd_rgbe_rle_bound(H, W));
        return GMD_ERR_INVALID;
    }
    if (W < 8 || W > 0x7fff) {  // not allowed to be run-length encoded: flat pixels
        memcpy(out, rgbe, (size_t)H * W * 4);
        *out_bytes = (int64_t)H * W * 4;
        return GMD_OK;
    }
    int64_t o = 0;
    uint8_t* plane = out + capacity - W;  // scratch for one component of one line: the last W bytes, behind the worst-case output
    for (int y = 0; y < H; ++y) {
        const uint8_t* line = rgbe + (int64_t)y * W * 4;
        out[o++] = 2; out[o++] = 2; out[o++] = (uint8_t)(W >> 8); out[o++] = (uint8_t)(W & 0xff);
        for (int c = 0; c < 4; ++c) {
            for (int x = 0; x < W; ++x) plane[x] = line[4 * x + c];
            o += rle_bytes(plane, W, out + o);
        }
    }
    *out_bytes = o;
    return GMD_OK;
}

}  // extern "C"
