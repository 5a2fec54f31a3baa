// Host-side preparation of the SMPL model image (HIP-free; shared by spsp_smpl_load and by the host
// emulation under tests/emul): blend-basis re-layout, joint-major LBS weights, Jbar/JS, tree levels,
// CSR of the extra joint regressors, and the packed fit constants.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "smpl_consts.hpp"

namespace spsp {

struct SmplHostImage {
    std::vector<float> basis, wT, jconst, fitc, csr_val;
    std::vector<int> level_joints, csr_ptr, csr_col;
    int parents[24];
    int level_start[26];
    int num_levels;
    int child_start[25];
    int child_list[24];
    int face_ids[kNFace];
    // Programs of the fit's tree WALKS (fit_core.hpp: fit_chain_walk, fit_reverse_walk_a / _b).  A walk is carried by QUADS of
    // lanes (quad = lane >> 2 holds one matrix row, lane & 3 its column); quad 3 f + i walks row i along leaf f's path.  Four
    // words per quad, then one word of children per joint (+ the padding joint 24):
    //   [0], [1]  the joints from below the root down to the leaf, 5 bits each (6 in the first word), padded with 24
    //   [2]       the leaf, then its ancestors as long as their subtree is a path, 5 bits each, padded with 24
    //   [3]       quads 0..2: the joints with a BRANCHING subtree from the deepest one up to the root, 5 bits each, padded with 24
    //   an idle quad / program has 31 in its first field
    //   walk[256 + k] = the three children of k, 5 bits each, 24 = none
    uint32_t walk[64 * 4 + 25];
    int walk_nb;             // joints with a branching subtree
    bool walk_ok;            // the tree fits that scheme (SMPL's does): joints with a branching subtree form one path from the root
};

// Returns an empty string on success, else the reason.
inline std::string smpl_prepare(const spsp_smpl_arrays& a, SmplHostImage& h, bool want_forward = true) {
    if (!(a.v_template && a.shapedirs && a.posedirs && a.J_regressor && a.lbs_weights && a.parents &&
          a.vertex_joint_ids && a.J_regressor_extra && a.J_regressor_cocoplus && a.J_regressor_h36m))
        return "NULL model array";
    if (a.parents[0] >= 0) return "parents[0] must be -1 (root)";
    for (int k = 1; k < kNJ; ++k)
        if (a.parents[k] < 0 || a.parents[k] >= k) {
            char buf[96];
            snprintf(buf, sizeof buf, "parents[%d]=%d is not an earlier joint", k, a.parents[k]);
            return buf;
        }
    for (int j = 0; j < 21; ++j)
        if (a.vertex_joint_ids[j] < 0 || a.vertex_joint_ids[j] >= kV) return "vertex_joint_ids out of range";

    if (want_forward) {
        // blend basis [217][20670]: rows 0..9 shapedirs^T, rows 10..216 posedirs
        h.basis.resize((size_t)kNCoef * kV3);
        for (int v = 0; v < kV; ++v)
            for (int c = 0; c < 3; ++c)
                for (int i = 0; i < kNB; ++i)
                    h.basis[(size_t)i * kV3 + 3 * v + c] = a.shapedirs[((size_t)v * 3 + c) * kNB + i];
        std::memcpy(h.basis.data() + (size_t)kNB * kV3, a.posedirs, sizeof(float) * (size_t)kNP * kV3);
        h.wT.resize((size_t)kNJ * kV);
        for (int v = 0; v < kV; ++v)
            for (int k = 0; k < kNJ; ++k) h.wT[(size_t)k * kV + v] = a.lbs_weights[(size_t)v * kNJ + k];
        // CSR of the 45 extra regressor rows (9 extra + 19 cocoplus + 17 h36m)
        const float* mats[3] = {a.J_regressor_extra, a.J_regressor_cocoplus, a.J_regressor_h36m};
        const int rows[3] = {9, 19, 17};
        h.csr_ptr.assign(1, 0);
        for (int m = 0; m < 3; ++m)
            for (int r = 0; r < rows[m]; ++r) {
                for (int v = 0; v < kV; ++v) {
                    const float w = mats[m][(size_t)r * kV + v];
                    if (w != 0.f) { h.csr_col.push_back(v); h.csr_val.push_back(w); }
                }
                h.csr_ptr.push_back((int)h.csr_col.size());
            }
        if (h.csr_col.empty()) { h.csr_col.push_back(0); h.csr_val.push_back(0.f); }
    }
    // Jbar = J_regressor @ v_template [24][3], JS[i][k*3+c] = (J_regressor @ shapedirs)[k][c][i]; fp64 accumulation
    h.jconst.assign(792, 0.f);
    {
        std::vector<double> acc(792, 0.0);
        for (int k = 0; k < kNJ; ++k)
            for (int v = 0; v < kV; ++v) {
                const double w = a.J_regressor[(size_t)k * kV + v];
                if (w == 0.0) continue;
                for (int c = 0; c < 3; ++c) {
                    acc[k * 3 + c] += w * a.v_template[3 * v + c];
                    for (int i = 0; i < kNB; ++i)
                        acc[72 + i * 72 + k * 3 + c] += w * a.shapedirs[((size_t)v * 3 + c) * kNB + i];
                }
            }
        for (int i = 0; i < 792; ++i) h.jconst[i] = (float)acc[i];
    }
    // tree: levels by depth, children lists
    {
        int depth[kNJ], maxd = 0;
        for (int k = 0; k < kNJ; ++k) {
            h.parents[k] = a.parents[k];
            depth[k] = k == 0 ? 0 : depth[a.parents[k]] + 1;
            maxd = std::max(maxd, depth[k]);
        }
        h.num_levels = maxd + 1;
        if (h.num_levels > 9) return "kinematic tree deeper than 9 levels (fit kernel limit; SMPL has 9)";
        for (int k = 0; k < kNJ; ++k) {
            int nc = 0;
            for (int c = 1; c < kNJ; ++c) nc += a.parents[c] == k;
            if (nc > 3) return "a joint with more than 3 children (fit kernel limit; SMPL has at most 3)";
        }
        h.level_joints.clear();
        for (int l = 0; l <= maxd; ++l) {
            h.level_start[l] = (int)h.level_joints.size();
            for (int k = 0; k < kNJ; ++k)
                if (depth[k] == l) h.level_joints.push_back(k);
        }
        for (int l = maxd + 1; l < 26; ++l) h.level_start[l] = (int)h.level_joints.size();
        for (int l = 0; l <= maxd; ++l)
            if (h.level_start[l + 1] - h.level_start[l] > 5) return "more than 5 joints on one tree level (fit kernel limit; SMPL has at most 5)";
        int n = 0;
        std::memset(h.child_list, 0, sizeof h.child_list);
        for (int k = 0; k < kNJ; ++k) {
            h.child_start[k] = n;
            for (int c = 1; c < kNJ; ++c)
                if (a.parents[c] == k) h.child_list[n++] = c;
        }
        h.child_start[24] = n;
        // walk programs
        int nch[kNJ], onlych[kNJ];
        bool is_path[kNJ];
        for (int k = 0; k < kNJ; ++k) { nch[k] = h.child_start[k + 1] - h.child_start[k]; onlych[k] = nch[k] ? h.child_list[h.child_start[k]] : -1; }
        for (int l = maxd; l >= 0; --l)
            for (int k = 0; k < kNJ; ++k)
                if (depth[k] == l) is_path[k] = nch[k] == 0 || (nch[k] == 1 && is_path[onlych[k]]);
        h.walk_ok = maxd <= 8;
        std::vector<int> bjoints, leaves;               // branching joints root first; leaves
        for (int cur = is_path[0] ? -1 : 0; cur >= 0;) {
            bjoints.push_back(cur);
            int nb = 0, next = -1;
            for (int c = h.child_start[cur]; c < h.child_start[cur + 1]; ++c)
                if (!is_path[h.child_list[c]]) { ++nb; next = h.child_list[c]; }
            if (nb > 1) h.walk_ok = false;
            cur = nb == 1 ? next : -1;
        }
        for (int k = 0; k < kNJ; ++k)
            if (nch[k] == 0) leaves.push_back(k);
        if (bjoints.size() > 6 || leaves.size() > 5) h.walk_ok = false;     // (<= 5 leaves: the 15 quads of a walk share wavefront 0)
        h.walk_nb = (int)bjoints.size();
        for (int k = 1; k < kNJ; ++k)
            if (a.parents[k] != kSmplParents[k]) h.walk_ok = false;      // the reverse walk's upper part is compiled for SMPL's tree (fit_core.hpp)
        for (int q = 0; q < 64 * 4; ++q) h.walk[q] = 0xffffffffu;
        auto put = [](uint32_t* w, const std::vector<int>& v, size_t first_word_fields) {
            uint32_t a0 = 0, a1 = 0;
            for (size_t j = 0; j < first_word_fields; ++j) a0 |= (uint32_t)(j < v.size() ? v[j] : 24) << (5 * j);
            for (size_t j = first_word_fields; j < first_word_fields + 2; ++j) a1 |= (uint32_t)(j < v.size() ? v[j] : 24) << (5 * (j - first_word_fields));
            w[0] = a0; w[1] = a1;
        };
        for (size_t f = 0; f < leaves.size() && f < 5; ++f) {
            std::vector<int> down, up;
            for (int j = leaves[f]; j > 0; j = a.parents[j]) down.insert(down.begin(), j);
            for (int j = leaves[f]; j >= 0 && is_path[j]; j = j ? a.parents[j] : -1) up.push_back(j);
            if (down.size() > 8 || up.size() > 6) h.walk_ok = false;
            for (int i = 0; i < 3; ++i) {
                uint32_t* w = h.walk + (3 * f + i) * 4;
                uint32_t tmp[2];
                put(w, down, 6);
                put(tmp, up, 6);
                w[2] = tmp[0];
            }
        }
        if (!bjoints.empty()) {
            std::vector<int> bup(bjoints.rbegin(), bjoints.rend());
            uint32_t tmp[2];
            put(tmp, bup, 6);
            for (int i = 0; i < 3; ++i) h.walk[i * 4 + 3] = tmp[0];
        }
        for (int k = 0; k <= kNJ; ++k) {
            uint32_t cw = 0;
            for (int n3 = 0; n3 < 3; ++n3) cw |= (uint32_t)(k < kNJ && n3 < nch[k] ? h.child_list[h.child_start[k] + n3] : 24) << (5 * n3);
            h.walk[256 + k] = cw;
        }
    }
    // fit constants (face vertices = first 5 picked ids: nose, reye, leye, rear, lear)
    h.fitc.assign(kFcTotal, 0.f);
    std::memcpy(h.fitc.data() + kFcJbar, h.jconst.data(), sizeof(float) * 72);
    for (int i = 0; i < kNB; ++i)
        std::memcpy(h.fitc.data() + kFcJS + i * kJSPitch, h.jconst.data() + 72 + i * 72, sizeof(float) * 72);
    for (int f = 0; f < kNFace; ++f) {
        const int v = a.vertex_joint_ids[f];
        h.face_ids[f] = v;
        for (int c = 0; c < 3; ++c) {
            h.fitc[kFcVt5 + f * 3 + c] = a.v_template[3 * v + c];
            for (int i = 0; i < kNB; ++i) h.fitc[kFcS5 + i * kF5Pitch + f * 3 + c] = a.shapedirs[((size_t)v * 3 + c) * kNB + i];
            for (int j = 0; j < kNP; ++j) h.fitc[kFcP5 + j * kF5Pitch + f * 3 + c] = a.posedirs[(size_t)j * kV3 + 3 * v + c];
        }
        for (int k = 0; k < kNJ; ++k) h.fitc[kFcW5 + f * 24 + k] = a.lbs_weights[(size_t)v * kNJ + k];
    }
    for (int i = 0; i < kNB; ++i) {
        for (int o = 0; o < 15; ++o) h.fitc[kFcC + i * kCPitch + o] = h.fitc[kFcS5 + i * kF5Pitch + o];
        for (int o = 0; o < 72; ++o) h.fitc[kFcC + i * kCPitch + 15 + o] = h.fitc[kFcJS + i * kJSPitch + o];
    }
    // relative rest joints J_k - J_parent(k) as affine functions of beta (root: J_0 itself)
    for (int k = 0; k < kNJ; ++k)
        for (int c = 0; c < 3; ++c) {
            const int o = k * 3 + c, po = a.parents[k] * 3 + c;
            h.fitc[kFcJrel + o] = h.jconst[o] - (k ? h.jconst[po] : 0.f);
            for (int i = 0; i < kNB; ++i)
                h.fitc[kFcJrel + 72 + i * kJSPitch + o] = h.jconst[72 + i * 72 + o] - (k ? h.jconst[72 + i * 72 + po] : 0.f);
        }
    // The pose feature is R[1:] - I; the kernel contracts posedirs with R directly, so the identity part
    // (-sum over the diagonal entries j % 9 in {0,4,8}) is folded into the template term once, in float64.
    for (int c = 0; c < 15; ++c) {
        double acc = h.fitc[kFcVt5 + c];
        for (int j = 0; j < kNP; ++j)
            if (j % 9 == 0 || j % 9 == 4 || j % 9 == 8) acc -= (double)h.fitc[kFcP5 + j * kF5Pitch + c];
        h.fitc[kFcVt5 + c] = (float)acc;
    }
    return "";
}

}  // namespace spsp
