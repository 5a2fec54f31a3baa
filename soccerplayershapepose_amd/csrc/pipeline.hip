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

}  // extern "C"
