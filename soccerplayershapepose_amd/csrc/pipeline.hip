// spsp_pipeline: the hot path end to end in ONE C call (the core of evaluate_model_2d, PR/PyTorch3DTest.py:3738-3776):
//   proxy -> regressor (cam, pose6d, betas) -> rot6d -> [PoseRelationModule] -> [fit] -> SMPL forward -> translation
//   [-> packed [cam | R | betas] record for the all-gather].
// Nothing new is computed here: it issues the launches of the stage entry points, in their order, on the caller's
// stream -- what player_recon.reconstruct_players does through six ctypes calls.  Every buffer is the caller's.
#include "smpl.hpp"

extern "C" {

int spsp_pipeline(spsp_ctx* ctx, spsp_regressor* regressor, spsp_smpl* smpl, spsp_relate* relate, const void* proxy,
                  int proxy_format, const float* boxes, const float* joints2d, const spsp_fit_config* fit, int B, float focal,
                  float resolution, const spsp_pipeline_out* out, void* stream) {
    SPSP_REQUIRE(ctx && regressor && smpl && out, "spsp_pipeline: NULL handle / output block");
    SPSP_REQUIRE(B >= 0, "spsp_pipeline: negative batch");
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(proxy && out->cam && out->pose6d && out->betas && out->rotmats, "spsp_pipeline: cam, pose6d, betas and rotmats outputs are required");
    SPSP_REQUIRE(!fit || joints2d, "spsp_pipeline: a fit configuration needs joints2d");
    SPSP_REQUIRE(!relate || (boxes && out->relate_tmp), "spsp_pipeline: the relation module needs boxes and relate_tmp [B,216]");
    int rc;
#define TRY(x) do { rc = (x); if (rc != SPSP_OK) return rc; } while (0)
    TRY(spsp_regressor_forward(ctx, regressor, proxy, proxy_format, B, out->cam, out->pose6d, out->betas, nullptr, stream));
    float* R = out->rotmats;
    if (fit && !relate) {
        // the fit kernel builds R from pose6d on its way in and writes the translation and the record on its way out: three
        // launches (and their boundaries) less in the tail of the chain, the same statements (geometry_device.hpp)
        spsp::FitEnds ends;
        ends.pose6d = out->pose6d;
        ends.translation = out->translation; ends.focal = focal; ends.resolution = resolution;
        ends.record = out->record;
        TRY(spsp::fit_launch(ctx, smpl, fit, R, out->betas, out->cam, joints2d, B, nullptr, nullptr, nullptr, ends, stream));
        if (out->vertices || out->joints) TRY(spsp_smpl_forward(ctx, smpl, R, out->betas, B, out->vertices, out->joints, stream));
        return SPSP_OK;
    }
    if (relate) {      // rot6d -> tmp, relation module tmp -> rotmats (it may not run in place)
        TRY(spsp_rot6d_to_rotmat(ctx, out->pose6d, B * 24, out->relate_tmp, stream));
        TRY(spsp_relate_forward(ctx, relate, out->relate_tmp, boxes, B, R, stream));
    } else {
        TRY(spsp_rot6d_to_rotmat(ctx, out->pose6d, B * 24, R, stream));
    }
    if (fit) TRY(spsp_fit(ctx, smpl, fit, R, out->betas, out->cam, joints2d, B, nullptr, nullptr, stream));
    if (out->vertices || out->joints) TRY(spsp_smpl_forward(ctx, smpl, R, out->betas, B, out->vertices, out->joints, stream));
    if (out->translation) TRY(spsp_wp_to_translation(ctx, out->cam, B, focal, resolution, out->translation, stream));
    if (out->record) TRY(spsp_pack_record(ctx, out->cam, R, out->betas, B, out->record, stream));
#undef TRY
    return SPSP_OK;
}

int spsp_pipeline_group(spsp_ctx* ctx, spsp_regressor* regressor, spsp_smpl* smpl, const void* const* proxies, int ngroups,
                        int proxy_format, const float* joints2d, const spsp_fit_config* fit, int B, float focal,
                        float resolution, const spsp_pipeline_out* out, void* stream) {
    SPSP_REQUIRE(ctx && regressor && smpl && out && proxies, "spsp_pipeline_group: NULL handle / output block");
    SPSP_REQUIRE(B >= 0 && ngroups >= 1 && ngroups <= 8, "spsp_pipeline_group: %d groups of %d", ngroups, B);
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(out->cam && out->pose6d && out->betas && out->rotmats, "spsp_pipeline_group: cam, pose6d, betas and rotmats outputs are required");
    SPSP_REQUIRE(!fit || joints2d, "spsp_pipeline_group: a fit configuration needs joints2d");
    const int N = ngroups * B;
    int rc;
#define TRY(x) do { rc = (x); if (rc != SPSP_OK) return rc; } while (0)
    for (int g = 0; g < ngroups; ++g) {
        SPSP_REQUIRE(proxies[g], "spsp_pipeline_group: NULL proxy batch %d", g);
        TRY(spsp_regressor_backbone(ctx, regressor, proxies[g], proxy_format, B, g * B, stream));
    }
    TRY(spsp_regressor_ief(ctx, regressor, N, out->cam, out->pose6d, out->betas, nullptr, stream));
    float* R = out->rotmats;
    if (fit) {
        spsp::FitEnds ends;
        ends.pose6d = out->pose6d;
        ends.translation = out->translation; ends.focal = focal; ends.resolution = resolution;
        ends.record = out->record;
        TRY(spsp::fit_launch(ctx, smpl, fit, R, out->betas, out->cam, joints2d, N, nullptr, nullptr, nullptr, ends, stream));
    } else {
        TRY(spsp_rot6d_to_rotmat(ctx, out->pose6d, N * 24, R, stream));
    }
    if (out->vertices || out->joints) TRY(spsp_smpl_forward(ctx, smpl, R, out->betas, N, out->vertices, out->joints, stream));
    if (!fit) {
        if (out->translation) TRY(spsp_wp_to_translation(ctx, out->cam, N, focal, resolution, out->translation, stream));
        if (out->record) TRY(spsp_pack_record(ctx, out->cam, R, out->betas, N, out->record, stream));
    }
#undef TRY
    return SPSP_OK;
}

}  // extern "C"
