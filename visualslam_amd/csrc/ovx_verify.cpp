// ovx_verify.cpp -- vxVerifyGraph: which node reads and writes what, the execution order that follows from it, the node fusions
// (ColorConvert into the pyramid's first launch, the detector's cell pass into the PyrLK launch, virtual grey images as level 0),
// user-kernel validators and metadata checks.  Host logic only.  Part of the OpenVX-subset runtime (ovx_internal.hpp).
#include "ovx_internal.hpp"

namespace ovxrt {

using namespace vsa;

// ---- dependency analysis -----------------------------------------------------------------------
// keys an object stands for: itself, plus -- for pyramids and their levels -- each other
void keys_of(_vx_reference *r, std::vector<const void *> *out)
{
    if (!valid(r)) return;
    out->push_back(r);
    if (r->type == VX_TYPE_PYRAMID) {
        for (auto *l : static_cast<_vx_pyramid *>(r)->levels) out->push_back(l);
    } else if (r->type == VX_TYPE_IMAGE) {
        auto *im = static_cast<_vx_image *>(r);
        if (im->pyramid) out->push_back(im->pyramid);
        for (_vx_image *p = im->parent; p; p = p->parent) out->push_back(p);     // a view stands for its parent too
    } else if (r->type == VX_TYPE_DELAY) {                                        // a delay stands for all of its slots
        for (auto *sl : static_cast<_vx_delay *>(r)->slots) keys_of(sl, out);
    }
}

bool intersects(const std::vector<const void *> &a, const std::vector<const void *> &b)
{
    for (auto x : a) for (auto y : b) if (x == y) return true;
    return false;
}

// the attribute of whatever object `r` is, through the public query of its type
vx_status query_any(_vx_reference *r, vx_enum attribute, void *ptr, vx_size size)
{
    switch (r->type) {
    case VX_TYPE_IMAGE: return vxQueryImage(static_cast<_vx_image *>(r), attribute, ptr, size);
    case VX_TYPE_ARRAY: return vxQueryArray(static_cast<_vx_array *>(r), attribute, ptr, size);
    case VX_TYPE_MATRIX: return vxQueryMatrix(static_cast<_vx_matrix *>(r), attribute, ptr, size);
    case VX_TYPE_SCALAR: return vxQueryScalar(static_cast<_vx_scalar *>(r), attribute, ptr, size);
    case VX_TYPE_PYRAMID: return vxQueryPyramid(static_cast<_vx_pyramid *>(r), attribute, ptr, size);
    case VX_TYPE_DELAY: return vxQueryDelay(static_cast<_vx_delay *>(r), attribute, ptr, size);
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

// a user node at vxVerifyGraph: the kernel's validator sees the parameters and describes every output in a meta
// format (homography_filter_node.cpp:162-202); what it set must agree with the object bound to that output
vx_status verify_user_node(_vx_node *nd)
{
    _vx_context *ctx = nd->ctx;
    _vx_kernel *k = nd->kernel;
    if (!k || !k->finalized) return VX_ERROR_INVALID_NODE;
    const vx_uint32 num = (vx_uint32)nd->params.size();
    std::vector<_vx_meta_format *> metas(num, nullptr);
    for (auto &m : metas) {
        m = new _vx_meta_format();
        m->type = VX_TYPE_META_FORMAT;
        m->ctx = ctx;
    }
    vx_status st = VX_SUCCESS;
    if (k->validate) st = k->validate(nd, nd->params.data(), num, metas.data());
    for (vx_uint32 i = 0; i < num && st == VX_SUCCESS; ++i) {      // OpenVX 1.0 validators: one call per parameter
        if (!nd->params[i]) continue;                               // an optional parameter left unset
        if (nd->spec[i].dir == VX_INPUT) { if (k->in_validate) st = k->in_validate(nd, i); }
        else if (k->out_validate) st = k->out_validate(nd, i, metas[i]);
    }
    if (st != VX_SUCCESS) {
        std::string msg = "vxVerifyGraph: the validator of user kernel '" + k->name + "' rejected its parameters";
        log_msg(ctx, nd, st, msg.c_str());
    }
    for (vx_uint32 i = 0; i < num && st == VX_SUCCESS; ++i) {
        _vx_reference *r = nd->params[i];
        if (!r || nd->spec[i].dir == VX_INPUT) continue;
        if (r->type == VX_TYPE_IMAGE) {    // a virtual image takes what the validator says
            auto *im = static_cast<_vx_image *>(r);
            auto w = metas[i]->attrs.find(VX_IMAGE_WIDTH), h = metas[i]->attrs.find(VX_IMAGE_HEIGHT);
            if (im->is_virtual && im->width == 0 && im->height == 0 && !im->dev && w != metas[i]->attrs.end() &&
                h != metas[i]->attrs.end() && w->second.size() == 4 && h->second.size() == 4) {
                memcpy(&im->width, w->second.data(), 4);
                memcpy(&im->height, h->second.data(), 4);
            }
        }
        for (auto &kv : metas[i]->attrs) {
            std::vector<uint8_t> have(kv.second.size());
            if (query_any(r, kv.first, have.data(), have.size()) != VX_SUCCESS || have != kv.second) {
                std::string msg = "vxVerifyGraph: an output of user kernel '" + k->name + "' does not match the validator's meta format";
                log_msg(ctx, nd, VX_ERROR_INVALID_PARAMETERS, msg.c_str());
                st = VX_ERROR_INVALID_PARAMETERS;
                break;
            }
        }
    }
    for (auto *m : metas) { m->magic = 0; delete m; }
    if (st != VX_SUCCESS) return st;
    if (k->init && !nd->user_initialized) {
        st = k->init(nd, nd->params.data(), num);
        if (st != VX_SUCCESS) return st;
    }
    nd->user_initialized = true;
    return VX_SUCCESS;
}

// The video stabilizer and the SFM sample convert the frame into a VIRTUAL grey image and build the pyramid from that
// (video_stabilizer/stabilizer.cpp:237-251, sfm/SfM.cpp:980-988) where the feature tracker converts straight into level 0
// (feature_tracker.cpp:360-362).  "Virtual" tells the runtime that the application never accesses the image, so the runtime
// gives it no memory of its own: it becomes a whole-image view of level 0 of the pyramid it feeds.  The GaussianPyramid node
// then has nothing to copy, and the same node fusions apply as in the feature tracker's graph (4 launches per frame instead
// of 7).  The view follows the pyramid object, so it stays right when a delay ages the pyramid's memory.
void alias_virtual_pyramid_inputs(_vx_graph *g)
{
    for (_vx_image *im : g->aliased)          // a re-verification starts over: node parameters may have changed
        if (valid(im) && im->parent) { release_ref(im->parent); im->parent = nullptr; im->dev = nullptr; im->pitch = 0; }
    g->aliased.clear();
    if (getenv("VISUALSLAM_AMD_NO_FUSION") != nullptr) return;
    for (_vx_node *gp : g->nodes) {
        if (gp->kind != K_GAUSSIAN_PYRAMID) continue;
        auto *gray = param<_vx_image>(gp, 0, VX_TYPE_IMAGE);
        auto *pyr = param<_vx_pyramid>(gp, 1, VX_TYPE_PYRAMID);
        if (!gray || !pyr || pyr->levels.empty()) continue;
        _vx_image *l0 = pyr->levels[0];
        if (gray == l0 || !gray->is_virtual || gray->parent || gray->dev || gray->pyramid || gray->has_fill) continue;
        if (gray->format != VX_DF_IMAGE_U8 || l0->format != VX_DF_IMAGE_U8) continue;
        const bool unsized = gray->width == 0 && gray->height == 0;
        if (!unsized && (gray->width != l0->width || gray->height != l0->height)) continue;
        // exactly one node writes the grey image, no other node writes the pyramid, and no other pyramid is built from it
        int gray_writers = 0;
        bool clash = false;
        for (_vx_node *nd : g->nodes)
            for (size_t p = 0; p < nd->params.size(); ++p) {
                _vx_reference *r = nd->params[p];
                if (!r || nd->spec[p].type == VX_TYPE_SCALAR) continue;
                const bool writes = nd->spec[p].dir != VX_INPUT;
                if (r == gray && writes) { ++gray_writers; if (nd->kind != K_COLOR_CONVERT && nd->kind != K_COPY_IMAGE) clash = true; }
                if (r == gray && !writes && nd->kind == K_GAUSSIAN_PYRAMID && nd != gp) clash = true;
                if (nd != gp && writes) {
                    std::vector<const void *> keys, mine{ pyr };
                    keys_of(r, &keys);
                    for (auto *l : pyr->levels) mine.push_back(l);
                    if (intersects(keys, mine)) clash = true;
                }
            }
        if (gray_writers != 1 || clash) continue;
        gray->width = l0->width;
        gray->height = l0->height;
        gray->parent = l0;
        gray->roi_x = gray->roi_y = 0;
        retain(l0);
        g->aliased.push_back(gray);
    }
}

vx_status verify_graph(_vx_graph *g)
{
    _vx_context *ctx = g->ctx;
    const size_t n = g->nodes.size();
    if (n == 0) return VX_ERROR_INVALID_GRAPH;
    // streams and pinned memory created below belong to the context's GPU, whatever device the calling thread last used
    // (a fresh host thread starts on device 0)
    if (!ctx->no_device && hipSetDevice(ctx->device) != hipSuccess) return VX_FAILURE;
    alias_virtual_pyramid_inputs(g);
    std::vector<std::vector<const void *>> rd(n), wr(n);
    for (size_t i = 0; i < n; ++i) {
        _vx_node *nd = g->nodes[i];
        for (size_t p = 0; p < nd->params.size(); ++p) {
            _vx_reference *r = nd->params[p];
            const ParamSpec &sp = nd->spec[p];
            if (!r) {
                if (!sp.optional) {
                    log_msg(ctx, nd, VX_ERROR_INVALID_PARAMETERS, "vxVerifyGraph: a required node parameter is missing");
                    return VX_ERROR_INVALID_PARAMETERS;
                }
                continue;
            }
            if (!valid(r, sp.type)) return VX_ERROR_INVALID_REFERENCE;
            if (sp.type == VX_TYPE_SCALAR) continue;
            if (sp.dir == VX_INPUT || sp.dir == VX_BIDIRECTIONAL) keys_of(r, &rd[i]);
            if (sp.dir == VX_OUTPUT || sp.dir == VX_BIDIRECTIONAL) {
                if (nd->kind == K_COLOR_CONVERT || nd->kind == K_COPY_IMAGE) {   // the image only (not the pyramid it may be level 0 of)
                    wr[i].push_back(r);
                    for (_vx_image *pp = static_cast<_vx_image *>(r)->parent; pp; pp = pp->parent) wr[i].push_back(pp);
                }
                else keys_of(r, &wr[i]);
            }
        }
    }
    // two writers of one object make the order ambiguous
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            std::vector<const void *> wi, wj;
            for (auto k : wr[i]) if (static_cast<const _vx_reference *>(k)->type != VX_TYPE_PYRAMID) wi.push_back(k);
            for (auto k : wr[j]) if (static_cast<const _vx_reference *>(k)->type != VX_TYPE_PYRAMID) wj.push_back(k);
            // ColorConvert writing level 0 and GaussianPyramid "owning" the pyramid is the one legal overlap
            const bool cvt_pyr = (g->nodes[i]->kind == K_GAUSSIAN_PYRAMID) != (g->nodes[j]->kind == K_GAUSSIAN_PYRAMID);
            if (!cvt_pyr && intersects(wi, wj)) {
                log_msg(ctx, g, VX_ERROR_INVALID_GRAPH, "vxVerifyGraph: two nodes write the same object");
                return VX_ERROR_INVALID_GRAPH;
            }
        }
    // Kahn's algorithm, ties broken by creation order.  A GaussianPyramid node whose input is its own
    // level 0 depends on the writer of that level, never on itself.
    std::vector<std::vector<size_t>> succ(n);
    std::vector<int> indeg(n, 0);
    for (size_t a = 0; a < n; ++a)
        for (size_t b = 0; b < n; ++b) {
            if (a == b) continue;
            if (intersects(wr[a], rd[b])) {
                succ[a].push_back(b);
                ++indeg[b];
            }
        }
    g->order.clear();
    std::vector<bool> done(n, false);
    for (size_t it = 0; it < n; ++it) {
        size_t pick = n;
        for (size_t i = 0; i < n; ++i) if (!done[i] && indeg[i] == 0) { pick = i; break; }
        if (pick == n) {
            log_msg(ctx, g, VX_ERROR_INVALID_GRAPH, "vxVerifyGraph: the graph has a cycle");
            return VX_ERROR_INVALID_GRAPH;
        }
        done[pick] = true;
        g->order.push_back(g->nodes[pick]);
        for (size_t s : succ[pick]) --indeg[s];
    }
    // Node fusion (what the callers' "-O3" verify option asks for, feature_tracker.cpp:435-436): ColorConvert writing level 0
    // of a pyramid that a GaussianPyramid node then builds from that same level -- the tracker graphs' first two nodes
    // (feature_tracker.cpp:360-362,386,393) -- run as ONE launch that reads the colour frame.  Legal when nothing reads
    // the grey image between the two nodes; the ColorConvert node then does nothing when executed.
    {
        for (_vx_node *nd : g->nodes) { nd->fused_away = false; nd->fused_cvt = nullptr; nd->fused_cells = nullptr; nd->cells_done = false;
                                         nd->pair = nullptr; nd->pair_leader = false; nd->pair_done = false; }
        std::vector<size_t> pos(n, 0);
        for (size_t k = 0; k < n; ++k)
            for (size_t i = 0; i < n; ++i)
                if (g->nodes[i] == g->order[k]) pos[i] = k;
        for (size_t gi = 0; gi < n && getenv("VISUALSLAM_AMD_NO_FUSION") == nullptr; ++gi) {
            _vx_node *gp = g->nodes[gi];
            if (gp->kind != K_GAUSSIAN_PYRAMID) continue;
            auto *gray = param<_vx_image>(gp, 0, VX_TYPE_IMAGE);
            auto *pyr = param<_vx_pyramid>(gp, 1, VX_TYPE_PYRAMID);
            if (!gray || !pyr || pyr->levels.size() < 2 || !is_level0(gray, pyr)) continue;
            for (size_t ci = 0; ci < n; ++ci) {
                _vx_node *cv = g->nodes[ci];
                if (cv->kind != K_COLOR_CONVERT || param<_vx_image>(cv, 1, VX_TYPE_IMAGE) != gray || pos[ci] > pos[gi]) continue;
                auto *frame = param<_vx_image>(cv, 0, VX_TYPE_IMAGE);
                if (!frame || frame->format != VX_DF_IMAGE_RGBX) continue;
                // every other reader of the grey image runs after the pyramid node, and -- the conversion now happens at the
                // pyramid node's place in the order -- no node between the two writes the colour frame or the grey image
                bool ok = true;
                std::vector<const void *> only_gray{ gray }, moved;
                keys_of(frame, &moved);
                keys_of(gray, &moved);
                for (size_t b = 0; b < n && ok; ++b) {
                    if (b == ci || b == gi) continue;
                    if (intersects(rd[b], only_gray) && pos[b] < pos[gi]) ok = false;
                    if (pos[b] > pos[ci] && pos[b] < pos[gi] && intersects(wr[b], moved)) ok = false;
                }
                if (ok) { cv->fused_away = true; gp->fused_cvt = cv; }
                break;
            }
        }
        // A HarrisTrack / FastTrack node that merges a PyrLK node's output (feature_tracker.cpp:402-430): only its merge pass needs the
        // tracked points; its cell pass reads the image and the mask, which are ready when PyrLK starts if no node from the
        // PyrLK node up to the HarrisTrack node writes them.  The cell pass then runs inside the PyrLK launch, whose last
        // waves otherwise leave compute units idle, and the frame has one launch less.
        for (size_t hi = 0; hi < n && getenv("VISUALSLAM_AMD_NO_FUSION") == nullptr; ++hi) {
            _vx_node *ht = g->nodes[hi];
            if (ht->kind != K_HARRIS_TRACK && ht->kind != K_FAST_TRACK) continue;
            auto *tracked = param<_vx_array>(ht, 3, VX_TYPE_ARRAY);
            if (!tracked) continue;
            for (size_t li = 0; li < n; ++li) {
                _vx_node *lk = g->nodes[li];
                if (lk->kind != K_PYRLK || lk->fused_cells || param<_vx_array>(lk, 4, VX_TYPE_ARRAY) != tracked || pos[li] > pos[hi]) continue;
                std::vector<const void *> inputs;
                keys_of(ht->params[0], &inputs);
                if (ht->params.size() > 2 && ht->params[2]) keys_of(ht->params[2], &inputs);
                bool ok = true;
                for (size_t b = 0; b < n && ok; ++b)
                    if (pos[b] >= pos[li] && pos[b] < pos[hi] && intersects(wr[b], inputs)) ok = false;
                if (ok) lk->fused_cells = ht;
                break;
            }
        }
        // A FindHomography node on the matches (stabilizer.cpp:264-273, SfM.cpp:1576-1578) and a detector node whose cell pass rides
        // in a PyrLK launch: what is left of the detector node, its merge pass, and the homography both read what PyrLK wrote and
        // neither reads what the other writes, so they run as ONE launch (homography.hip) at the place of whichever comes first.
        // Legal when the later node can move up: it does not depend on the earlier one, and no node between the two writes what
        // it reads, or reads or writes what it writes.
        for (size_t fi = 0; fi < n && getenv("VISUALSLAM_AMD_NO_FUSION") == nullptr; ++fi) {
            _vx_node *fh = g->nodes[fi];
            if (fh->kind != K_FIND_HOMOGRAPHY || fh->pair) continue;
            for (size_t hi = 0; hi < n; ++hi) {
                _vx_node *ht = g->nodes[hi];
                if ((ht->kind != K_HARRIS_TRACK && ht->kind != K_FAST_TRACK) || ht->pair) continue;
                size_t li = n;
                for (size_t q = 0; q < n; ++q) if (g->nodes[q]->kind == K_PYRLK && g->nodes[q]->fused_cells == ht) li = q;
                const size_t a = pos[fi] < pos[hi] ? fi : hi, b = a == fi ? hi : fi;      // a leads, b moves up to a's place
                if (li == n || pos[li] > pos[a]) continue;
                bool ok = !intersects(wr[a], rd[b]) && !intersects(wr[b], rd[a]) && !intersects(wr[a], wr[b]);
                for (size_t c = 0; c < n && ok; ++c)
                    if (pos[c] > pos[a] && pos[c] < pos[b] &&
                        (intersects(wr[c], rd[b]) || intersects(wr[b], rd[c]) || intersects(wr[b], wr[c]))) ok = false;
                if (!ok) continue;
                fh->pair = ht; ht->pair = fh;
                g->nodes[a]->pair_leader = true;
                break;
            }
        }
    }
    // virtual images created with width = height = 0 take their size from the node that writes them
    for (_vx_node *nd : g->order) {
        if (nd->kind != K_COLOR_CONVERT && nd->kind != K_COPY_IMAGE) continue;
        auto *in = param<_vx_image>(nd, 0, VX_TYPE_IMAGE), *out = param<_vx_image>(nd, 1, VX_TYPE_IMAGE);
        if (in && out && out->is_virtual && out->width == 0 && out->height == 0 && !out->dev) {
            out->width = in->width;
            out->height = in->height;
        }
    }
    // metadata checks that need no device
    for (_vx_node *nd : g->order) {
        if (nd->kind == K_USER) {
            vx_status us = verify_user_node(nd);
            if (us != VX_SUCCESS) return us;
        }
        if (nd->kind == K_WARP_PERSPECTIVE) {
            auto *in = param<_vx_image>(nd, 0, VX_TYPE_IMAGE), *out = param<_vx_image>(nd, 3, VX_TYPE_IMAGE);
            auto *m = param<_vx_matrix>(nd, 1, VX_TYPE_MATRIX);
            auto *t = param<_vx_scalar>(nd, 2, VX_TYPE_SCALAR);
            if (!in || !out || in->format != out->format) {
                log_msg(ctx, nd, VX_ERROR_INVALID_FORMAT, "vxWarpPerspectiveNode: input and output must both be U8 or both RGBX");
                return VX_ERROR_INVALID_FORMAT;
            }
            if (out->width == 0 || out->height == 0) return VX_ERROR_INVALID_DIMENSION;
            if (!m || m->data_type != VX_TYPE_FLOAT32 || m->cols != 3 || m->rows != 3) return VX_ERROR_INVALID_DIMENSION;
            if (!t || (t->v.e != VX_INTERPOLATION_BILINEAR && t->v.e != VX_INTERPOLATION_NEAREST_NEIGHBOR)) {
                log_msg(ctx, nd, VX_ERROR_NOT_SUPPORTED, "vxWarpPerspectiveNode: NEAREST_NEIGHBOR and BILINEAR are offered");
                return VX_ERROR_NOT_SUPPORTED;
            }
        }
    }
    for (_vx_node *nd : g->nodes) {
        if (nd->kind == K_FIND_HOMOGRAPHY) {
            auto *a = param<_vx_array>(nd, 0, VX_TYPE_ARRAY), *b = param<_vx_array>(nd, 1, VX_TYPE_ARRAY);
            auto *m = param<_vx_matrix>(nd, 2, VX_TYPE_MATRIX);
            auto *k = param<_vx_array>(nd, 9, VX_TYPE_ARRAY);
            auto *meth = param<_vx_scalar>(nd, 3, VX_TYPE_SCALAR);
            if (!a || !b || a->item_type != NVX_TYPE_POINT2F || b->item_type != NVX_TYPE_POINT2F || (k && k->item_type != VX_TYPE_UINT8)) {
                log_msg(ctx, nd, VX_ERROR_INVALID_TYPE, "nvxFindHomographyNode: points must be NVX_TYPE_POINT2F arrays, the mask a VX_TYPE_UINT8 array");
                return VX_ERROR_INVALID_TYPE;
            }
            if (!m || m->data_type != VX_TYPE_FLOAT32 || m->cols != 3 || m->rows != 3) return VX_ERROR_INVALID_DIMENSION;
            if (!meth || meth->v.e != NVX_FIND_HOMOGRAPHY_METHOD_RANSAC) {
                log_msg(ctx, nd, VX_ERROR_NOT_SUPPORTED, "nvxFindHomographyNode: only NVX_FIND_HOMOGRAPHY_METHOD_RANSAC is implemented");
                return VX_ERROR_NOT_SUPPORTED;
            }
            if (!(scalar_f32(nd, 4) > 0.0f)) return VX_ERROR_INVALID_VALUE;
        }
        if (nd->kind == K_PYRLK) {
            auto *a = param<_vx_array>(nd, 2, VX_TYPE_ARRAY), *c = param<_vx_array>(nd, 4, VX_TYPE_ARRAY);
            ItemKind k;
            if (!a || !c || a->item_type != c->item_type || !item_kind(a->item_type, &k)) {
                log_msg(ctx, nd, VX_ERROR_NOT_SUPPORTED, "vxOpticalFlowPyrLKNode: point arrays must both be NVX_TYPE_POINT2F or NVX_TYPE_KEYPOINTF");
                return VX_ERROR_NOT_SUPPORTED;
            }
            if (pyrlk_window(nd) < (vx_size)kLkMinWin || pyrlk_window(nd) > (vx_size)kLkMaxWin) return VX_ERROR_INVALID_VALUE;
        }
        if (nd->kind == K_HARRIS_TRACK || nd->kind == K_FAST_TRACK) {
            auto *o = param<_vx_array>(nd, 1, VX_TYPE_ARRAY);
            ItemKind k;
            if (!o || !item_kind(o->item_type, &k)) return VX_ERROR_NOT_SUPPORTED;
            if (scalar_u32(nd, 6) == 0) return VX_ERROR_INVALID_VALUE;
        }
    }
    // bookkeeping for the post-run read-backs
    g->written.clear();
    g->corner_nodes.clear();
    g->matrices.clear();
    for (_vx_node *nd : g->order) {
        for (size_t p = 0; p < nd->params.size(); ++p)
            if (nd->kind != K_USER && nd->params[p] && nd->spec[p].type == VX_TYPE_MATRIX && nd->spec[p].dir != VX_INPUT) {
                auto *m = static_cast<_vx_matrix *>(nd->params[p]);
                if (std::find(g->matrices.begin(), g->matrices.end(), m) == g->matrices.end()) g->matrices.push_back(m);
            }
        for (size_t p = 0; p < nd->params.size(); ++p)
            if (nd->params[p] && nd->spec[p].type == VX_TYPE_ARRAY && nd->spec[p].dir != VX_INPUT) {
                auto *a = static_cast<_vx_array *>(nd->params[p]);
                if (std::find(g->written.begin(), g->written.end(), a) == g->written.end()) g->written.push_back(a);
            }
        if ((nd->kind == K_HARRIS_TRACK || nd->kind == K_FAST_TRACK) && param<_vx_scalar>(nd, 7, VX_TYPE_SCALAR))
            g->corner_nodes.push_back(nd);
    }
    if (!ctx->no_device) {
        size_t slots = 2 * (g->written.size() + g->corner_nodes.size() + 1);   // 8 bytes per slot
        for (_vx_matrix *m : g->matrices) slots += 2 * ((m->host.size() + 7) / 8);
        if (slots > g->pinned_slots) {
            if (g->pinned) (void)hipHostFree(g->pinned);
            g->pinned = nullptr;
            if (hipHostMalloc(reinterpret_cast<void **>(&g->pinned), slots * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
                return VX_ERROR_NO_MEMORY;
            g->pinned_slots = slots;
        }
        if (!g->stream) {
            // VISUALSLAM_AMD_STREAM_PRIORITY=<p> (-1 greatest, 1 least; unset = normal) creates the graphs' streams with that
            // priority.  HIP keeps a pool of hardware queues per priority and deals a pool's streams round-robin over its queues:
            // at normal priority the graphs share the pool with the process's other streams (32 camera streams landed 9/7/7/9 on
            // the four queues), with a priority of their own they get 8/8/8/8 -- +2.8 % on the 32-stream bench, the same for -1
            // and 1 (tools/queue_probe.hip, profiles/r02_kernel_experiments.txt).  Opt-in: a priority also ranks the graphs'
            // kernels against the application's other GPU work.
            const char *e = getenv("VISUALSLAM_AMD_STREAM_PRIORITY");
            const hipError_t err = e ? hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, atoi(e))
                                     : hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
            if (err != hipSuccess) return VX_FAILURE;
        }
    }
    g->verified = true;
    g->state = VX_GRAPH_STATE_VERIFIED;
    return VX_SUCCESS;
}

// does any node of the graph write `r` (or memory `r` stands for: its pyramid, its ROI parents, views of it)?
bool written_in_graph(_vx_graph *g, _vx_reference *r)
{
    if (!g || !r) return false;
    std::vector<const void *> mine, theirs;
    keys_of(r, &mine);
    for (_vx_node *nd : g->nodes)
        for (size_t p = 0; p < nd->params.size(); ++p) {
            if (!nd->params[p] || nd->spec[p].dir == VX_INPUT || nd->spec[p].type == VX_TYPE_SCALAR) continue;
            theirs.clear();
            keys_of(nd->params[p], &theirs);
            if (intersects(mine, theirs)) return true;
        }
    return false;
}

}  // namespace ovxrt
