/*
 * train_example.c -- the whole training path through the C ABI (include/rlhip.h) from plain C: what a JNI / cgo / N-API binding does,
 * without the binding.  Trains LambdaMART on a tiny synthetic set and prints RankLib's model text.
 *
 *   gcc -std=c99 -I include integration/c/train_example.c -L ranklib_amd/lib -lrlhip -Wl,-rpath,$PWD/ranklib_amd/lib -o train_example
 *
 * (tests/test_abi.py compiles this file with -fsyntax-only: the header must stay valid C99.)
 */
#include <stdio.h>
#include <stdlib.h>
#include "rlhip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != RL_OK) { fprintf(stderr, "%s: %s (%d)\n", #call, rl_last_error(), rc_); return 1; } } while (0)

int main(void)
{
    enum { Q = 40, PER = 12, N = Q * PER, F = 5, ROUNDS = 10 };
    float *X = (float *)malloc(sizeof(float) * N * F), *lab = (float *)malloc(sizeof(float) * N);
    int32_t qoff[Q + 1];
    unsigned s = 12345u;
    for (int q = 0; q <= Q; q++) qoff[q] = q * PER;
    for (int i = 0; i < N; i++) {
        float z = 0.f;
        for (int f = 0; f < F; f++) { s = s * 1664525u + 1013904223u; X[i * F + f] = (float)(s >> 8) / 16777216.f; z += (f < 2) ? X[i * F + f] : 0.f; }
        lab[i] = (float)(int)(z * 2.f);                        /* relevance 0 .. 3 */
    }
    rl_params p;
    rl_params_default(&p);
    p.n_trees = ROUNDS; p.n_leaves = 6;
    rl_trainer *t = NULL;
    CHECK(rl_create(&p, &t));
    /* rows in blocks, as a caller with a bounded staging buffer does (rl_set_rows); X != NULL in rl_set_train is the one-shot form */
    CHECK(rl_set_train(t, NULL, N, F, lab, qoff, Q, NULL, NULL));
    CHECK(rl_set_rows(t, 0, 0, N / 2, X));
    CHECK(rl_set_rows(t, 0, N / 2, N - N / 2, X + (size_t)(N / 2) * F));
    CHECK(rl_init(t));
    for (int m = 0; m < ROUNDS; m++) {
        float tm = 0.f, vm = 0.f; int32_t stop = 0;
        CHECK(rl_boost_round(t, NULL, &tm, &vm, &stop));
        printf("%4d  NDCG@10-T %.4f\n", m + 1, tm);
    }
    double train_score = 0.0;
    CHECK(rl_finish(t, &train_score, NULL));
    int64_t need = 0;
    CHECK(rl_model_to_text(t, NULL, 0, &need));
    char *text = (char *)malloc((size_t)need);
    CHECK(rl_model_to_text(t, text, need, &need));
    printf("NDCG@10 on training data: %.4f\n%s", train_score, text);
    free(text); free(X); free(lab);
    rl_destroy(t);
    return 0;
}
