// rl_model.h -- host-side tree / ensemble representation and RankLib's model text format.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace rl {

// one regression tree, nodes in pre-order (root, left subtree, right subtree)
struct HostTree {
    int n_nodes = 0;
    float weight = 0.f;                 // Ensemble.weights[i]   learning/tree/Ensemble.java:34
    std::vector<int32_t> feature;       // feature ID, -1 = leaf
    std::vector<float> threshold;
    std::vector<int32_t> left, right;
    std::vector<float> output;
    std::vector<double> deviance;
    std::vector<int32_t> count;
};

struct ModelHeader {                    // the "## ..." lines of LambdaMART.model()  LambdaMART.java:292-297
    int n_trees, n_leaves, n_threshold;
    float learning_rate;
    int early_stop;
    const char *name = "LambdaMART";     // Ranker.name(): "LambdaMART" or "MART" (learning/tree/MART.java:41-44)
};

// java.lang.Float.toString / Double.toString renderings (shortest decimal that round-trips, Java's
// decimal/scientific switch at 1e-3 and 1e7).
std::string java_float_to_string(float v);
std::string java_double_to_string(double v);

std::string model_to_text(const ModelHeader &h, const std::vector<HostTree> &trees);
// Parses what LambdaMART.loadFromString accepts ("##" lines dropped, element order
// feature/threshold/left/right: parsing/ModelLineProducer.java:43-78, learning/tree/Ensemble.java:45-70,141-159).
// Returns false and sets err on malformed input.
bool model_from_text(const std::string &text, std::vector<HostTree> &trees, std::string &err);

}  // namespace rl
