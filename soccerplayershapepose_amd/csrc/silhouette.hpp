// The rasteriser handle shared by silhouette.hip (coverage image + metric) and silhouette_grad.hip (ownership map, loss, backward).
#pragma once
#include <vector>

#include "common.hpp"

struct spsp_rasterizer {
    spsp::DevicePool pool;
    int* faces;              // device [F][3]
    std::vector<int> host_faces;
    int F, max_batch, wh;
    unsigned int* mask;      // [max_batch][(2 wh)^2 / 32]
    float* partial;          // [max_batch][ceil(wh^2 / 256)]
    // differentiable path (silhouette_grad.hip), allocated at first use
    float* proj = nullptr;   // [max_batch][6890] float4 (x_n, y_n, z, -)
    int* owner = nullptr;    // [max_batch][(2 wh)^2] front-most face copy per sample, -1 = empty
    float* gface = nullptr;  // [max_batch][F][3][2]
    float* tpart = nullptr;  // block partials of dL/dt
    float* vpart = nullptr;  // block partials of the loss sum
    int* adj_ptr = nullptr;  // vertex -> incident face corners (CSR)
    int* adj_idx = nullptr;
};

namespace spsp {
int rasterizer_grad_workspace(spsp_rasterizer* h);
int silhouette_loss_grad_impl(spsp_ctx* ctx, spsp_rasterizer* h, const float* verts, const float* translation, int tstride, float focal, int B,
                              const uint8_t* target, int target_shared, const float* weight, int wstride, float* sil_sum, int sstride,
                              float* grad_verts, float* grad_translation, int gstride, float* image, void* stream);
}
