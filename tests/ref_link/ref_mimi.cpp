// Build-container-only integration check (needs /root/reference): the REFERENCE's own Mimi codec graph builders
// (src/moshi/models/compression.h, modules/{conv,seanet,transformer}.h, quantization/{vq,core_vq}.h - compiled from where they
// lie, nothing copied) linked against this repository's ggml surface, over the SAME weight tensors the moshi_hot driver created.
// mimi_decode (codes -> 1920 samples) and mimi_encode (1920 samples -> codes) are run frame by frame next to
// moshi_hot_mimi_decode / moshi_hot_mimi_encode on the same executor (CPU oracle on the host device): outputs must be bit-identical.
#include <assert.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <deque>
#include <iostream>
#include <map>
#include <string>
#include <vector>
#include <moshi/ptrs.h>
#include <moshi/safetensor.h>
#include <ggml.h>
#include <ggml-backend.h>
#include <ggml-cpu.h>
#define CAPTURE(...)
#define CAPTURE_GROUP(...)
#define ONCE(code) {static bool once=false; if (!once) {{code;}; once=true;}}
#define ON_NTH(nth, code) {static int count=0; if (count++ == (nth)) {code;}}
#include "src/context.h"
#include "src/loader.h"
#include "src/torch.h"
#include "src/moshi/modules/transformer.h"
#include "src/moshi/quantization/core_vq.h"
#include "src/moshi/quantization/vq.h"
#include "src/moshi/modules/conv.h"
#include "src/moshi/modules/seanet.h"
#include "src/moshi/models/compression.h"
#include "moshi_hot.h"

static moshi_hot_model_t * model;
static ggml_tensor * W(const std::string & name, bool optional = false) {
    ggml_tensor * t = moshi_hot_weight(model, name.c_str());
    if (!t && !optional) { fprintf(stderr, "missing weight %s\n", name.c_str()); exit(2); }
    return t;
}
static moshi_streaming_conv_1d_t * conv(const std::string & n, int cin, int cout, int k, int s, bool bias = true) {
    return new moshi_streaming_conv_1d_t{ cin, cout, k, s, W(n + ".weight"), bias ? W(n + ".bias") : NULL };
}
static moshi_streaming_conv_transpose_1d_t * convtr(const std::string & n, int cin, int cout, int k, int s, int groups, bool bias) {
    return new moshi_streaming_conv_transpose_1d_t{ cin, cout, k, s, groups, W(n + ".weight"), bias ? W(n + ".bias") : NULL };
}
static moshi_seanet_resnet_block_t * res(const std::string & n, int dim) {
    auto r = new moshi_seanet_resnet_block_t;
    r->block_1 = conv(n + ".block.1.conv", dim, dim / 2, 3, 1);
    r->block_3 = new moshi_stateless_conv_1d_t{ dim / 2, dim, 1, W(n + ".block.3.conv.weight"), W(n + ".block.3.conv.bias") };
    return r;
}
static moshi_rvq_t * rvq(const std::string & n, int n_q) {
    auto r = new moshi_rvq_t;
    r->n_q = n_q;
    r->vq = new moshi_residual_vq_t;
    for (int i = 0; i < n_q; i++) {
        auto vq = new moshi_vq_t;
        vq->_codebook = new moshi_EuclideanCodebook_t{ W(n + ".vq.layers." + std::to_string(i) + "._codebook.embedding") };
        r->vq->layers.push_back(vq);
    }
    r->output_proj = new torch_nn_conv1d_t{ W(n + ".output_proj.weight") };
    r->input_proj = new torch_nn_conv1d_t{ W(n + ".input_proj.weight") };
    return r;
}
static moshi_streaming_transformer_t * mimi_transformer(const std::string & n) {
    auto tr = new moshi_streaming_transformer_t;
    tr->context = 250; tr->weights_per_step = 0; tr->capacity = 250; tr->rope_max_period = 10000; tr->dim_per_head = 64;
    for (int l = 0; l < 8; l++) {
        const std::string p = n + ".layers." + std::to_string(l) + ".";
        auto layer = new moshi_streaming_transformer_layer_t;
        layer->norm1 = new torch_nn_layer_norm_t{ 1e-5f, W(p + "norm1.weight"), W(p + "norm1.bias") };
        auto attn = new moshi_smha_t;
        attn->embed_dim = 512; attn->num_heads = 8; attn->cross_attention = false; attn->cache_cross_attention = false;
        attn->causal = true; attn->rope_max_period = 10000; attn->context = 250; attn->weights_per_step = 0;
        attn->in_projs.push_back(new torch_nn_linear_t{ W(p + "self_attn.in_projs.weight"), NULL });
        attn->out_projs.push_back(new torch_nn_linear_t{ W(p + "self_attn.out_projs.weight"), NULL });
        layer->self_attn = attn;
        layer->layer_scale_1 = new moshi_layer_scale_t{ W(p + "layer_scale_1.scale") };
        layer->norm2 = new torch_nn_layer_norm_t{ 1e-5f, W(p + "norm2.weight"), W(p + "norm2.bias") };
        layer->linear1 = new torch_nn_linear_t{ W(p + "linear1.weight"), NULL };
        layer->linear2 = new torch_nn_linear_t{ W(p + "linear2.weight"), NULL };
        layer->layer_scale_2 = new moshi_layer_scale_t{ W(p + "layer_scale_2.scale") };
        tr->layers.push_back(layer);
    }
    return tr;
}

int main(int argc, char ** argv) {
    const char * oracle_path = argc > 1 ? argv[1] : "oracle/liboracle.so";
    const int frames = argc > 2 ? atoi(argv[2]) : 6;
    void * h = dlopen(oracle_path, RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", oracle_path, dlerror()); return 2; }
    void * fn = dlsym(h, "oracle_graph_compute");
    if (!fn) { fprintf(stderr, "oracle_graph_compute not found\n"); return 2; }
    ggml_backend_cpu_set_graph_compute((ggml_backend_cpu_graph_compute_t) fn);
    ggml_backend_load_all();
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, NULL);
    assert(cpu);

    moshi_hot_config cfg;
    moshi_hot_config_moshika(&cfg);
    cfg.enable_lm = 0;
    cfg.mimi_n_q = 4; cfg.mimi_codebook_size = 64;
    model = moshi_hot_create(cpu, &cfg, 0);
    assert(model);

    auto mimi = new moshi_mimi_t;
    mimi->initialized = true; mimi->frame_rate = 12.5f; mimi->sample_rate = 24000;
    mimi->quantizer = new moshi_split_rvq_t;
    mimi->quantizer->n_q_semantic = 1;
    mimi->quantizer->rvq_first = rvq("mimi.quantizer.rvq_first", 1);
    mimi->quantizer->rvq_rest = rvq("mimi.quantizer.rvq_rest", cfg.mimi_n_q - 1);
    mimi->upsample = convtr("mimi.upsample.convtr", 512, 512, 4, 2, 512, false);
    mimi->decoder_transformer = mimi_transformer("mimi.decoder_transformer.transformer");
    {
        auto d = new moshi_seanet_decoder_t;
        d->model_0 = conv("mimi.decoder.model.0.conv", 512, 1024, 7, 1);
        d->model_2 = convtr("mimi.decoder.model.2.convtr", 1024, 512, 16, 8, 1, true);  d->model_3 = res("mimi.decoder.model.3", 512);
        d->model_5 = convtr("mimi.decoder.model.5.convtr", 512, 256, 12, 6, 1, true);   d->model_6 = res("mimi.decoder.model.6", 256);
        d->model_8 = convtr("mimi.decoder.model.8.convtr", 256, 128, 10, 5, 1, true);   d->model_9 = res("mimi.decoder.model.9", 128);
        d->model_11 = convtr("mimi.decoder.model.11.convtr", 128, 64, 8, 4, 1, true);   d->model_12 = res("mimi.decoder.model.12", 64);
        d->model_14 = conv("mimi.decoder.model.14.conv", 64, 1, 3, 1);
        mimi->decoder = d;
    }
    mimi->downsample = conv("mimi.downsample.conv", 512, 512, 4, 2, false);
    mimi->encoder_transformer = mimi_transformer("mimi.encoder_transformer.transformer");
    {
        auto e = new moshi_seanet_encoder_t;
        e->model_0 = conv("mimi.encoder.model.0.conv", 1, 64, 7, 1);
        e->model_1 = res("mimi.encoder.model.1", 64);    e->model_3 = conv("mimi.encoder.model.3.conv", 64, 128, 8, 4);
        e->model_4 = res("mimi.encoder.model.4", 128);   e->model_6 = conv("mimi.encoder.model.6.conv", 128, 256, 10, 5);
        e->model_7 = res("mimi.encoder.model.7", 256);   e->model_9 = conv("mimi.encoder.model.9.conv", 256, 512, 12, 6);
        e->model_10 = res("mimi.encoder.model.10", 512); e->model_12 = conv("mimi.encoder.model.12.conv", 512, 1024, 16, 8);
        e->model_14 = conv("mimi.encoder.model.14.conv", 1024, 512, 3, 1);
        mimi->encoder = e;
    }

    ScratchContext scratch(256, cpu);
    StateContext dec_state(cpu), enc_state(cpu);
    NE upsample_ne = { 1, 512, 1, 1 }, decoder_ne = { 2, 512, 1, 1 };   // src/moshi.cpp:240-241
    auto dstates = moshi_mimi_states(&dec_state, mimi, upsample_ne, decoder_ne);
    dec_state.alloc(); dec_state.init();
    init(&scratch, dstates, mimi);
    auto estates = moshi_mimi_encoder_states(&enc_state, mimi);
    enc_state.alloc(); enc_state.init();
    init(&scratch, estates, mimi);

    uint64_t rng = 777;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    int bad = 0;
    for (int f = 0; f < frames; f++) {
        // decode
        std::vector<int> codes((size_t) cfg.mimi_n_q);
        int32_t codes32[32];
        for (int i = 0; i < cfg.mimi_n_q; i++) { codes[(size_t) i] = (int) (next() % (uint64_t) cfg.mimi_codebook_size); codes32[i] = codes[(size_t) i]; }
        std::vector<float> ref_pcm, pcm(1920);
        mimi_decode(scratch, mimi, dstates, codes, ref_pcm);
        moshi_hot_mimi_decode(model, codes32, pcm.data());
        int diff = ref_pcm.size() != 1920;
        for (size_t i = 0; i < 1920 && i < ref_pcm.size(); i++) if (memcmp(&ref_pcm[i], &pcm[i], 4) != 0) diff++;
        if (diff) { bad++; fprintf(stderr, "frame %d decode: %d of 1920 samples differ (ref size %zu; first %.9g vs %.9g)\n", f, diff, ref_pcm.size(), ref_pcm.empty() ? 0.f : ref_pcm[0], pcm[0]); }
        // encode
        std::vector<float> frame(1920);
        for (auto & v : frame) v = ((float) (next() % 20001) / 10000.f - 1.f) * 0.3f;
        std::vector<int> ref_codes;
        int32_t got_codes[32] = { 0 };
        mimi_encode(scratch, mimi, estates, frame, ref_codes);
        moshi_hot_mimi_encode(model, frame.data(), got_codes);
        int cdiff = (int) ref_codes.size() != cfg.mimi_n_q;
        for (size_t i = 0; i < ref_codes.size() && i < 32; i++) if (ref_codes[i] != got_codes[i]) cdiff++;
        if (cdiff) { bad++; fprintf(stderr, "frame %d encode: codes differ (ref n=%zu):", f, ref_codes.size()); for (size_t i = 0; i < ref_codes.size(); i++) fprintf(stderr, " %d/%d", ref_codes[i], got_codes[i]); fprintf(stderr, "\n"); }
    }
    if (bad) {   // where do the two encoder graphs part ways?
        ggml_cgraph * ga = estates->encoder_graph.ctx->gf, * gb = moshi_hot_graph(model, 2);
        const int na = ggml_graph_n_nodes(ga), nb = ggml_graph_n_nodes(gb);
        fprintf(stderr, "encoder graphs: reference %d nodes, moshi_hot %d nodes\n", na, nb);
        for (int i = 0; i < na && i < nb; i++) {
            ggml_tensor * a = ggml_graph_node(ga, i), * b = ggml_graph_node(gb, i);
            const bool same = a->op == b->op && a->type == b->type && a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3] &&
                              a->nb[1] == b->nb[1] && a->nb[2] == b->nb[2] && memcmp(a->op_params, b->op_params, 32) == 0;
            if (!same) {
                fprintf(stderr, "first structural difference at node %d: reference %s [%lld %lld %lld %lld] type %d params %d %d | moshi_hot %s [%lld %lld %lld %lld] type %d params %d %d\n", i,
                        ggml_op_name(a->op), (long long) a->ne[0], (long long) a->ne[1], (long long) a->ne[2], (long long) a->ne[3], a->type, a->op_params[0], a->op_params[1],
                        ggml_op_name(b->op), (long long) b->ne[0], (long long) b->ne[1], (long long) b->ne[2], (long long) b->ne[3], b->type, b->op_params[0], b->op_params[1]);
                break;
            }
        }
    }
    printf("reference mimi builders vs moshi_hot: %d frames decode + encode, %d mismatches\n", frames, bad);
    moshi_hot_free(model);
    return bad ? 1 : 0;
}
