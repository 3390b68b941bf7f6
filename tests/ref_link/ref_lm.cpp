// Build-container-only integration check (needs /root/reference): the REFERENCE's own LM frame driver and graph builders -
// src/moshi/models/lm.h (moshi_lmgen_step :778-979, the delay ring :715-743, the embedding sums :555-607, the text head :659-677, the chained
// depformer graph :446-553), lm_utils.h, utils/sampling.h, modules/transformer.h, torch.h, context.h - compiled from where they lie, nothing
// copied, are linked against this repository's ggml surface (include/*.h + libggml-mi355x.so) and stepped over the SAME weight tensors the
// moshi_hot driver created, with the same inputs. Both run on the same executor (the CPU oracle attached to the host device), so if
// moshi_hot.cpp restates lm.h faithfully the tokens a caller gets back are identical frame after frame: return flag, delayed text token, delayed
// audio tokens - for moshika (dep_q 8) and PersonaPlex (dep_q 16 chained, 8 exposed, delay ring one row deeper: lm.h:727-729, 802-805), greedy and in
// the reference's sampling mode with the same host rand() stream (src/context.h:465-480).
//
// include/moshi/moshi.h is needed for the plain struct `Entry` only; it includes <sentencepiece_processor.h> (not in this image) for its tokenizer
// API, so the test puts an EMPTY header of that name on the include path (tests/ref_link/stub/). Nothing of SentencePiece is declared or used.
//
//   g++ -std=c++20 -Itests/ref_link/stub -I<repo>/include -I/root/reference/include -I/root/reference ref_lm.cpp -L<repo>/moshi.cpp_amd -lmoshi-hot -lggml-mi355x -ldl
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
#include <moshi/moshi.h>
#define CAPTURE(...)
#define CAPTURE_GROUP(...)
#define ONCE(code) {static bool once=false; if (!once) {{code;}; once=true;}}
#define ON_NTH(nth, code) {static int count=0; if (count++ == (nth)) {code;}}
#include "src/context.h"
#include "src/loader.h"
#include "src/torch.h"
#include "src/moshi/modules/transformer.h"
#include "src/moshi/utils/sampling.h"
#include "src/moshi/models/lm_utils.h"
#include "src/moshi/models/lm.h"
#include "moshi_hot.h"

static ggml_tensor * W(moshi_hot_model_t * m, const std::string & name) {
    ggml_tensor * t = moshi_hot_weight(m, name.c_str());
    if (!t) { fprintf(stderr, "missing weight %s\n", name.c_str()); exit(2); }
    return t;
}
static torch_nn_linear_t * lin(moshi_hot_model_t * m, const std::string & name) { return new torch_nn_linear_t{ W(m, name), NULL }; }

// moshi_streaming_transformer_t over the driver's tensors, built the way lm_default.h:18-146 builds it (RMS norms, gated FFN, one weight set or one per step)
static moshi_streaming_transformer_t * make_transformer(moshi_hot_model_t * m, const std::string & name, int dim, int heads, int layers, int context, int max_period, int n_sets) {
    auto tr = new moshi_streaming_transformer_t;
    tr->context = context; tr->weights_per_step = 0; tr->capacity = context; tr->rope_max_period = max_period; tr->dim_per_head = dim / heads;
    for (int l = 0; l < layers; l++) {
        const std::string p = name + ".layers." + std::to_string(l) + ".";
        auto layer = new moshi_streaming_transformer_layer_t;
        layer->norm1_rms = new moshi_rms_norm_t{ 1e-8f, W(m, p + "norm1.alpha") };
        auto attn = new moshi_smha_t;
        attn->embed_dim = dim; attn->num_heads = heads; attn->cross_attention = false; attn->cache_cross_attention = n_sets > 1;
        attn->causal = true; attn->rope_max_period = max_period; attn->context = context; attn->weights_per_step = 0;
        layer->norm2_rms = new moshi_rms_norm_t{ 1e-8f, W(m, p + "norm2.alpha") };
        for (int k = 0; k < n_sets; k++) {
            const std::string ws = n_sets > 1 ? "." + std::to_string(k) : "";
            attn->in_projs.push_back(lin(m, p + "self_attn.in_projs" + ws + ".weight"));
            attn->out_projs.push_back(lin(m, p + "self_attn.out_projs" + ws + ".weight"));
            auto gating = new moshi_activation_gating_t;
            gating->linear_in = lin(m, p + "gating" + ws + ".linear_in.weight");
            gating->linear_out = lin(m, p + "gating" + ws + ".linear_out.weight");
            layer->gating.push_back(gating);
        }
        layer->self_attn = attn;
        tr->layers.push_back(layer);
    }
    return tr;
}

struct frame_out { int ok, text; std::vector<int> audio; };

int main(int argc, char ** argv) {
    const char * oracle_path = argc > 1 ? argv[1] : "oracle/liboracle.so";
    const int steps = argc > 2 ? atoi(argv[2]) : 40;
    const bool personaplex = argc > 3 && atoi(argv[3]) != 0;
    const bool sampled = argc > 4 && atoi(argv[4]) != 0;
    void * h = dlopen(oracle_path, RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", oracle_path, dlerror()); return 2; }
    void * fn = dlsym(h, "oracle_graph_compute");
    if (!fn) { fprintf(stderr, "oracle_graph_compute not found\n"); return 2; }
    ggml_backend_cpu_set_graph_compute((ggml_backend_cpu_graph_compute_t) fn);
    ggml_backend_load_all();
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, NULL);
    assert(cpu);

    // the driver's model: tools/moshi-config.json / personaplex-config.json structure (17 codebooks, their delays, dep_q 8 / 16, Depth ring of 8) at small widths
    moshi_hot_config cfg;
    if (personaplex) moshi_hot_config_personaplex(&cfg); else moshi_hot_config_moshika(&cfg);
    cfg.dim = 512; cfg.num_heads = 4; cfg.num_layers = 2; cfg.ffn_hidden = 768; cfg.context = 24;   // ring of 24 < steps: the Temporal ring wraps
    cfg.text_card = 500; cfg.card = 64;
    cfg.dep_dim = 256; cfg.dep_heads = 4; cfg.dep_layers = 2; cfg.dep_ffn_hidden = 512;
    cfg.mimi_n_q = 8; cfg.mimi_codebook_size = 64;
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0;
    if (sampled) { cfg.temp = 0.8f; cfg.temp_text = 0.7f; cfg.top_k = 250; cfg.top_k_text = 25; }   // tools/moshi-sts.cpp:106-107
    else { cfg.temp = 0.f; cfg.temp_text = 0.f; }
    moshi_hot_model_t * model = moshi_hot_create(cpu, &cfg, 0);
    assert(model);
    const int exposed = personaplex ? 8 : cfg.dep_q;                  // lm.h:802-805
    const int needed = cfg.n_q + 1 - exposed - 1;                     // the other speaker's codebooks a caller passes in

    // ---- the reference's model struct over the driver's weight tensors (lm_default.h:18-227)
    auto lm = new moshi_lmmodel_t;
    lm->n_q = cfg.n_q; lm->dep_q = cfg.dep_q; lm->card = cfg.card; lm->text_card = cfg.text_card; lm->dim = cfg.dim;
    lm->max_delay = 0;
    for (int i = 0; i <= cfg.n_q; i++) { lm->delays.push_back(cfg.delays[i]); if (cfg.delays[i] > lm->max_delay) lm->max_delay = cfg.delays[i]; }
    lm->demux_second_stream = false;
    lm->text_emb = new moshi_scaled_embedding_t{ NULL, W(model, "lm.text_emb.weight") };
    for (int k = 0; k < cfg.n_q; k++) lm->emb.push_back(new moshi_scaled_embedding_t{ NULL, W(model, "lm.emb." + std::to_string(k) + ".weight") });
    lm->text_linear = lin(model, "lm.text_linear.weight");
    lm->transformer = make_transformer(model, "lm.transformer", cfg.dim, cfg.num_heads, cfg.num_layers, cfg.context, cfg.max_period, 1);
    lm->out_norm = new moshi_rms_norm_t{ 1e-8f, W(model, "lm.out_norm.alpha") };
    lm->depformer_multi_linear = true;
    for (int k = 0; k < cfg.dep_q; k++) {
        lm->depformer_in.push_back(lin(model, "lm.depformer_in." + std::to_string(k) + ".weight"));
        lm->linears.push_back(lin(model, "lm.linears." + std::to_string(k) + ".weight"));
        if (k > 0) lm->depformer_emb.push_back(new moshi_scaled_embedding_t{ NULL, W(model, "lm.depformer_emb." + std::to_string(k - 1) + ".weight") });
    }
    lm->depformer_text_emb = new moshi_scaled_embedding_t{ NULL, W(model, "lm.depformer_text_emb.weight") };
    lm->depformer = make_transformer(model, "lm.depformer", cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_context, 0, cfg.dep_q);
    lm->num_codebooks = cfg.n_q + 1; lm->num_audio_codebooks = cfg.n_q; lm->audio_offset = 1; lm->delay_steps = 0;
    lm->text_initial_token_id = cfg.text_card; lm->initial_token_id = cfg.card;   // lm_default.h:218-221
    lm->personaplex = personaplex;

    moshi_lmgen_t gen;
    gen.lm = lm; gen.use_sampling = sampled; gen.temp = cfg.temp; gen.temp_text = cfg.temp_text; gen.top_k = cfg.top_k; gen.top_k_text = cfg.top_k_text;
    gen.machine = NULL; gen.machine_state = NULL; gen.condition_sum = NULL; gen.text_prefixes = NULL; gen.audio_prefixes = NULL;

    StateContext state_ctx(cpu);
    auto lm_states = moshi_lmmodel_states(&state_ctx, lm, NULL);
    state_ctx.alloc();
    state_ctx.init();
    ScratchContext scratch(256, cpu);
    init(&scratch, lm_states, lm, NULL);
    auto gen_state = moshi_lmgen_state(lm);

    // ---- the same pseudo-random stream of the other speaker's codes for both; the host noise stream (rand()) is restarted in front of each pass
    std::vector<std::vector<int>> inputs((size_t) steps, std::vector<int>((size_t) needed));
    uint64_t rng = 12345;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (auto & f : inputs) for (auto & t : f) t = (int) (next() % (uint64_t) cfg.card);

    std::vector<frame_out> a, b;
    srand(777);
    for (int step = 0; step < steps; step++) {           // the driver (include/moshi_hot.h: moshi_hot_lm_step = one moshi_lmgen_step)
        int32_t in_audio[32] = { 0 }, txt = -7, aud[64];
        for (int i = 0; i < 64; i++) aud[i] = -7;
        for (int i = 0; i < needed; i++) in_audio[i] = inputs[(size_t) step][(size_t) i];
        frame_out o;
        o.ok = moshi_hot_lm_step(model, in_audio, &txt, aud);
        o.text = txt; o.audio.assign(aud, aud + exposed);
        a.push_back(o);
    }
    srand(getenv("REF_LM_OTHER_NOISE") ? 778 : 777);     // (REF_LM_OTHER_NOISE: sensitivity check - another noise stream must change sampled tokens)
    for (int step = 0; step < steps; step++) {           // the reference
        int txt = -7;
        std::vector<int> aud(inputs[(size_t) step]);
        frame_out o;
        o.ok = moshi_lmgen_step(scratch, &gen, gen_state, lm_states, false, txt, aud) ? 1 : 0;
        o.text = txt; o.audio = aud; o.audio.resize((size_t) exposed, -7);
        b.push_back(o);
    }
    int bad = 0, valid = 0;
    std::map<int, int> seen;
    for (auto & f : a) if (f.ok) { seen[f.text]++; for (int t : f.audio) seen[1000000 + t]++; }
    for (int step = 0; step < steps; step++) {
        const frame_out & x = a[(size_t) step], & y = b[(size_t) step];
        bool same = x.ok == y.ok;
        if (same && x.ok) { valid++; same = x.text == y.text && x.audio == y.audio; }
        if (!same) {
            bad++;
            fprintf(stderr, "frame %d: driver ok %d text %d audio", step, x.ok, x.text);
            for (int t : x.audio) fprintf(stderr, " %d", t);
            fprintf(stderr, " | reference ok %d text %d audio", y.ok, y.text);
            for (int t : y.audio) fprintf(stderr, " %d", t);
            fprintf(stderr, "\n");
        }
    }
    // the two ran on the same executor from the same weights: identical state must have produced identical rings too (the Depth ring wraps inside one
    // PersonaPlex frame; the Temporal ring wrapped at frame 24)
    printf("reference lm.h (moshi_lmgen_step) vs moshi_hot: %s %s, %d frames (%d with output, %d distinct token values), %d mismatching frames\n", personaplex ? "personaplex" : "moshika",
           sampled ? "sampled" : "greedy", steps, valid, (int) seen.size(), bad);
    moshi_hot_free(model);
    return bad ? 1 : 0;
}
