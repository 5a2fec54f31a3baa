// Implicit-GEMM convolution on gfx950 MFMA -- interface between regressor.hip (layer plan) and conv.hip.
#pragma once
#include "common.hpp"

namespace spsp {

// All activations are bf16 NHWC with a ZERO BORDER baked into the buffer (3 px for the network input,
// 1 px for every feature map).  The border makes every filter tap of every output pixel an in-bounds
// read, so the gather has no masks: the address of tap-step `s` of output pixel m is
//     in + in_off + row_base(m) + step_off(s),      row_base(m) = b*in_img + ho*stride*in_row + wo*stride*in_pix
// and one K-step is a CONTIGUOUS run of BK bf16 values of the input:
//   3x3 / 1x1 convs (Cin >= 64): step = (tap r, tap c, 64-channel chunk ch)  -> step_off = r*in_row + c*in_pix + ch*64
//   conv1 7x7/2 (Cin = 18):      step = kernel row r; the 7 taps x 18 channels of that row are 126 contiguous
//                                values, padded to BK = 128 with two zero weights -> step_off = r*in_row
struct ConvArgs {
    const uint16_t* in;
    const uint16_t* w;       // [N][nsteps*BK] bf16, K ordered like the steps
    const float* bias;       // [N] folded BN bias
    const uint16_t* res;     // residual, addressed like `out` (may be null)
    uint16_t* out;
    float* pool_out;         // if non-null: [B][N] fp32 mean over the image's Ho*Wo pixels (fused global avg-pool;
                             // requires BM == Ho*Wo so one M-tile is one image); `out` is then not written
    int in_img, in_row, in_pix, in_off;
    int out_img, out_row, out_pix, out_off;
    int Ho, Wo, M, N;
    int stride;
    int kw_steps, chunks, nsteps;
    int relu;
    float* part;             // unused since round 2 (split-K over workgroups is gone)
    int splitk;              // 1; the stamped twin library uses -1 as a diagnosis switch (conv_patch.hip)
    unsigned int* counter;   // conv_wres: eight device words, 128 bytes apart, zero before the launch -- dynamic tile scheduler
    int f16;                 // element type of in / w / res / out: 0 = bf16, 1 = fp16 (conv_elem.hpp)
    int xcd_n;               // conv_patch: XCDs along the channel blocks (tile order; set by launch_conv_patch)
    // conv_ring: an optional SECOND convolution by the same workgroups -- the 1x1 stride-2 downsample of a residual stage rides
    // with the stage's 3x3 stride-2 conv: same input pixels (the 3x3's centre tap), same output geometry and channel count;
    // `chunks` extra K-steps into a second accumulator set.  w2 = [n2][chunks*64], no residual, no ReLU.
    const uint16_t* w2;
    const float* bias2;
    uint16_t* out2;
    int n2;
    unsigned long long* dbg; // diagnostic builds (-DSPSP_CONV_STAMPS): s_memtime stamps of workgroup 0
};

// second generation (conv_ring.hip): LDS-DMA ring, BK = 64, optional split-K
enum RingTile { kRing128x128s3 = 0, kRing128x128s2, kRing128x64s3, kRing128x64s4, kRing64x128s3, kRing64x64s4, kRingNone };
int launch_conv_ring(const ConvArgs& a, RingTile tile, hipStream_t stream);

// third generation (conv_patch.hip): 3x3 stride-1 convs with an LDS-resident input patch, 256 / 128 px x 64 / 32 ch per workgroup
bool conv_patch_supported(const ConvArgs& a);
int launch_conv_patch(const ConvArgs& a, bool two_patch_buffers, int tile_pixels, int ring_groups, int block_channels, bool producer_waves, hipStream_t stream);

int conv_patch_xcd_n(const ConvArgs& a, int tile_pixels, int block_channels = 64);   // XCDs along the channel blocks for the fewest fabric bytes

// conv_p4.hip: the same convolutions on 256-pixel tiles without the split of K over wave pairs: four accumulating waves over
// the whole K + four producer waves (layer2)
bool conv_p4_supported(const ConvArgs& a);
int launch_conv_p4(const ConvArgs& a, int max_workgroups, hipStream_t stream);      // fewer workgroups than tiles: persistent

// conv_s2.hip: the 3x3 stride-2 conv of a residual stage with the stage's 1x1 stride-2 downsample riding on it (w2 / bias2 /
// out2 / n2), on an LDS-resident patch split into the four parity planes of the input; 128 output pixels x 64 channels per workgroup
bool conv_s2_supported(const ConvArgs& a);
int launch_conv_s2(const ConvArgs& a, hipStream_t stream);

// conv_s2rs.hip: layer2.0's 3x3 / 2 + 1x1 / 2 pair (one 64-channel chunk of input) with the weights stationary in registers and
// two tiles per workgroup requested up front
bool conv_s2rs_supported(const ConvArgs& a);
int launch_conv_s2rs(const ConvArgs& a, int num_cus, hipStream_t stream);

// conv_s2q.hip (round 5): layer2.0's pair with ALL 128 output channels in one workgroup (planes staged once), wave = a 32-channel
// quarter over the whole K, weights in registers (AGPR-operand MFMAs), 4 x 16-pixel tiles, no K split.  in [B][66][66][64], out / out2
// [B][34][34][128]; wf = pack_s2q_fragments' image on the device
void pack_s2q_fragments(const uint16_t* w, const uint16_t* w2, std::vector<uint16_t>& out);
int launch_conv_s2q(const uint16_t* in, const uint16_t* wf, const float* bias, const float* bias2, uint16_t* out, uint16_t* out2, int B,
                    int f16, int relu, int max_workgroups, hipStream_t stream);

// conv_rs.hip (round 5, plan "throughput"): layer2's 3x3 / 1 convolutions (128 -> 128, 32 x 32 maps) with a 64-channel block's weights
// stationary in registers, persistent over 8 x 16-pixel tiles.  in / out / res [B][34][34][128]; wf = pack_rs_fragments' image
void pack_rs_fragments(const uint16_t* w, std::vector<uint16_t>& out);
int launch_conv_rs(const uint16_t* in, const uint16_t* wf, const float* bias, const uint16_t* res, uint16_t* out, int B, int f16, int relu,
                   int max_workgroups, hipStream_t stream);

// conv_wres.hip: 3x3 stride-1 64->64 convs (layer1) with all nine weight slabs resident in LDS, persistent workgroups
bool conv_wres_supported(const ConvArgs& a);
int launch_conv_wres(const ConvArgs& a, int num_cus, hipStream_t stream);

// conv_fb.hip (round 5): one launch per BasicBlock of layer1 -- conv1 + ReLU into LDS, conv2 + residual + ReLU out of it, both
// convolutions' weights stationary in registers.  in / out [B][66][66][64] (distinct); wf = pack_block_fragments' image on the device
void pack_block_fragments(const uint16_t* w1, const uint16_t* w2, std::vector<uint16_t>& out);
int launch_conv_fb(const uint16_t* in, const uint16_t* wf, const float* bias1, const float* bias2, uint16_t* out, int B, int f16,
                   int max_workgroups, hipStream_t stream);

// conv_stem.hip: conv1 + BN + ReLU + max-pool in one kernel: native input [B][262][264][18] -> pooled map [B][66][66][64];
// its first workgroup also zeroes `nzero` words at `zero_words` (the counters of the forward's later kernels; may be NULL)
int launch_conv_stem(const uint16_t* in, const uint16_t* w, const float* bias, uint16_t* out, int B, int f16, int num_cus,
                     unsigned int* zero_words, int nzero, hipStream_t stream, int rows_per_band = 0, int mfma32 = 0,
                     float* zero_wide = nullptr, int nzero_wide16 = 0);   // zero_wide: 16-byte words every workgroup zeroes a share of

}  // namespace spsp
