// Build-container-only integration check (needs /root/reference): the REFERENCE's own graph builders
// (src/moshi/modules/transformer.h, rope.h, gating.h, src/torch.h, src/context.h - compiled from where they lie, nothing copied)
// are linked against this repository's ggml surface (include/*.h + libggml-mi355x.so) and build + run the Temporal transformer
// stack over the SAME weight tensors the moshi_hot driver created. Both graphs run on the same executor (the CPU oracle attached
// to the host device), so if moshi_hot.cpp restates the reference's graph construction faithfully the two outputs are bit-identical,
// step after step (KV ring, RoPE offset, bias-mask window and ring indices included).
//
//   g++ -std=c++20 -I<repo>/include -I/root/reference/include -I/root/reference ref_transformer.cpp -L<repo>/moshi.cpp_amd -lggml-mi355x -ldl
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
#include "moshi_hot.h"

static ggml_tensor * W(moshi_hot_model_t * m, const std::string & name) {
    ggml_tensor * t = moshi_hot_weight(m, name.c_str());
    if (!t) { fprintf(stderr, "missing weight %s\n", name.c_str()); exit(2); }
    return t;
}

int main(int argc, char ** argv) {
    const char * oracle_path = argc > 1 ? argv[1] : "oracle/liboracle.so";
    const int steps = argc > 2 ? atoi(argv[2]) : 40;
    void * h = dlopen(oracle_path, RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", oracle_path, dlerror()); return 2; }
    void * fn = dlsym(h, "oracle_graph_compute");
    if (!fn) { fprintf(stderr, "oracle_graph_compute not found\n"); return 2; }
    ggml_backend_cpu_set_graph_compute((ggml_backend_cpu_graph_compute_t) fn);
    ggml_backend_load_all();
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, NULL);
    assert(cpu);

    // the driver's model: same small shape family as tests (dim 512, 4 heads of 128, gated FFN 768, ring of 24 < steps: it wraps)
    moshi_hot_config cfg;
    moshi_hot_config_moshika(&cfg);
    cfg.dim = 512; cfg.num_heads = 4; cfg.num_layers = 2; cfg.ffn_hidden = 768; cfg.context = 24;
    cfg.text_card = 500; cfg.card = 64; cfg.n_q = 6; cfg.dep_q = 3;
    const int delays[7] = { 0, 0, 1, 1, 0, 1, 1 };
    for (int i = 0; i < MOSHI_HOT_MAX_CODEBOOKS; i++) cfg.delays[i] = i < 7 ? delays[i] : 0;
    cfg.dep_dim = 256; cfg.dep_heads = 4; cfg.dep_layers = 2; cfg.dep_ffn_hidden = 512; cfg.dep_context = 3;
    cfg.mimi_n_q = 3; cfg.mimi_codebook_size = 64;
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0;
    moshi_hot_model_t * model = moshi_hot_create(cpu, &cfg, 0);
    assert(model);

    // the reference's structs over the driver's weight tensors
    auto tr = new moshi_streaming_transformer_t;
    tr->context = cfg.context; tr->weights_per_step = 0; tr->capacity = cfg.context; tr->rope_max_period = cfg.max_period;
    tr->dim_per_head = cfg.dim / cfg.num_heads;
    for (int l = 0; l < cfg.num_layers; l++) {
        const std::string p = "lm.transformer.layers." + std::to_string(l) + ".";
        auto layer = new moshi_streaming_transformer_layer_t;
        layer->norm1_rms = new moshi_rms_norm_t{ 1e-8f, W(model, p + "norm1.alpha") };
        auto attn = new moshi_smha_t;
        attn->embed_dim = cfg.dim; attn->num_heads = cfg.num_heads; attn->cross_attention = false; attn->cache_cross_attention = false;
        attn->causal = true; attn->rope_max_period = cfg.max_period; attn->context = cfg.context; attn->weights_per_step = 0;
        attn->in_projs.push_back(new torch_nn_linear_t{ W(model, p + "self_attn.in_projs.weight"), NULL });
        attn->out_projs.push_back(new torch_nn_linear_t{ W(model, p + "self_attn.out_projs.weight"), NULL });
        layer->self_attn = attn;
        layer->norm2_rms = new moshi_rms_norm_t{ 1e-8f, W(model, p + "norm2.alpha") };
        auto gating = new moshi_activation_gating_t;
        gating->linear_in = new torch_nn_linear_t{ W(model, p + "gating.linear_in.weight"), NULL };
        gating->linear_out = new torch_nn_linear_t{ W(model, p + "gating.linear_out.weight"), NULL };
        layer->gating.push_back(gating);
        tr->layers.push_back(layer);
    }
    StateContext state_ctx(cpu);
    auto states = moshi_streaming_transformer_state(&state_ctx, tr, NULL);
    state_ctx.alloc();
    state_ctx.init();
    ScratchContext scratch(256, cpu);
    init(&scratch, states, tr, NULL);

    uint64_t rng = 12345;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    double max_in = 0;
    int bad = 0;
    for (int step = 0; step < steps; step++) {
        int32_t in_audio[32] = { 0 }, txt = 0, aud[32] = { 0 };
        for (int i = 0; i < cfg.n_q - cfg.dep_q; i++) in_audio[i] = (int32_t) (next() % (uint64_t) cfg.card);
        moshi_hot_lm_step(model, in_audio, &txt, aud);
        std::vector<float> xin((size_t) cfg.dim), want((size_t) cfg.dim), got((size_t) cfg.dim);
        if (moshi_hot_read_last(model, "transformer_in", xin.data(), cfg.dim) || moshi_hot_read_last(model, "stack_out", want.data(), cfg.dim)) { fprintf(stderr, "read_last failed\n"); return 2; }
        auto x = scratch.input(GGML_NE(cfg.dim, 1), xin);
        auto r = moshi_streaming_transformer_graph(scratch, tr, states, x);
        ggml_backend_tensor_get(r, got.data(), 0, (size_t) cfg.dim * 4);
        int diff = 0;
        for (int i = 0; i < cfg.dim; i++) { if (memcmp(&want[(size_t) i], &got[(size_t) i], 4) != 0) diff++; max_in = fmax(max_in, fabs((double) xin[(size_t) i])); }
        if (diff) { bad++; fprintf(stderr, "step %d: %d of %d outputs differ (first want %.9g got %.9g)\n", step, diff, cfg.dim, want[0], got[0]); }
    }
    printf("reference transformer builders vs moshi_hot: %d steps, ring %d, %d mismatching steps (max |input| %.3g)\n", steps, cfg.context, bad, max_in);
    moshi_hot_free(model);
    return bad ? 1 : 0;
}
