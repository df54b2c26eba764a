// C++ mirror of the reference's doctests / src/main.rs on top of the C ABI (include/friedrich.hpp).
// The reference only prints; the expected values here are the CPU oracle's (tests/golden/golden_v1.json,
// "readme_default" and "readme_1d"), which this program receives on its command line from tests/test_gpu_cpp.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "friedrich.hpp"

using namespace friedrich;

static int failures = 0;
#define EXPECT(cond, ...)                                \
    do {                                                 \
        if (!(cond)) {                                   \
            ++failures;                                  \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                    \
            std::printf("\n");                           \
        }                                                \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 6) {
        std::printf("usage: %s iterations predict_1 variance_1 noise prior\n", argv[0]);
        return 2;
    }
    const int want_iter = std::atoi(argv[1]);
    const double want_pred = std::atof(argv[2]), want_var = std::atof(argv[3]), want_noise = std::atof(argv[4]),
                 want_prior = std::atof(argv[5]);

    // src/main.rs:14-27 / mod.rs doctests: GaussianProcess::default on the 4-point set
    const std::vector<std::vector<double>> training_inputs = {{0.8}, {1.2}, {3.8}, {4.2}};
    const std::vector<double> training_outputs = {3.0, 4.0, -2.0, -2.0};
    auto gp = GaussianProcess<>::default_(training_inputs, training_outputs);
    const std::vector<double> input = {1.0};  // a Vec<f64> is ONE sample (conversion/mod.rs:95-118)
    const double mean = gp.predict(input)[0];
    const double var = gp.predict_variance(input)[0];
    std::printf("prediction: %.9f +- %.9f   (iterations %zu, noise %.9f, prior %.9f)\n", mean, std::sqrt(var),
                gp.last_fit_iterations(), gp.noise, gp.prior.c);
    EXPECT((int)gp.last_fit_iterations() == want_iter, "iterations %zu != %d", gp.last_fit_iterations(), want_iter);
    EXPECT(std::fabs(mean - want_pred) < 1e-7, "predict %.12f != %.12f", mean, want_pred);
    EXPECT(std::fabs(var - want_var) < 1e-7, "variance %.12f != %.12f", var, want_var);
    EXPECT(std::fabs(gp.noise - want_noise) < 1e-8, "noise %.12f != %.12f", gp.noise, want_noise);
    EXPECT(std::fabs(gp.prior.c - want_prior) < 1e-12, "prior %.12f != %.12f", gp.prior.c, want_prior);
    auto mv = gp.predict_mean_variance(input);
    EXPECT(std::fabs(mv.first[0] - mean) < 1e-10 && std::fabs(mv.second[0] - var) < 1e-10, "predict_mean_variance disagrees");
    const double lik = gp.likelihood();
    std::printf("likelihood of the current model : %.9f\n", lik);
    EXPECT(std::isfinite(lik), "likelihood not finite");

    // src/main.rs:29-41: add_samples + fit_parameters
    const std::vector<std::vector<double>> additional_inputs = {{0.0}, {1.0}, {2.0}, {5.0}};
    const std::vector<double> additional_outputs = {2.0, 3.0, -1.0, -2.0};
    gp.add_samples(additional_inputs, additional_outputs);
    EXPECT(gp.nb_samples() == 8, "nb_samples %lld", (long long)gp.nb_samples());
    gp.fit_parameters(true, true, 100, 0.05, std::chrono::seconds(3600));
    const std::vector<std::vector<double>> inputs = {{1.0}, {2.0}, {3.0}};
    const auto outputs = gp.predict(inputs);
    std::printf("predictions: %.6f %.6f %.6f\n", outputs[0], outputs[1], outputs[2]);
    EXPECT(outputs.size() == 3 && outputs[0] > outputs[1] && outputs[1] > outputs[2], "predictions not decreasing");

    // src/main.rs:43-51: sample_at + sample
    const std::vector<std::vector<double>> new_inputs = {{1.0}, {2.0}};
    const auto sampler = gp.sample_at(new_inputs);
    std::mt19937_64 rng(42);
    for (int i = 1; i <= 3; ++i) {
        const auto s = sampler.sample(rng);
        std::printf("sample %d : %.6f %.6f\n", i, s[0], s[1]);
        EXPECT(s.size() == 2 && std::isfinite(s[0]) && std::isfinite(s[1]), "sample not finite");
    }
    EXPECT(sampler.cholesky_covariance(0, 1) == 0.0, "unpack() must zero the upper triangle");
    EXPECT(std::fabs(sampler.mean()[0] - outputs[0]) < 1e-9, "sampler mean != predict");

    // builder.rs doctest: explicit kernel / prior / noise (src/gaussian_process/mod.rs:106-127)
    {
        auto gp2 = GaussianProcess<>::builder(training_inputs, training_outputs)
                       .set_noise(0.1)
                       .set_kernel(Exponential())
                       .set_prior(LinearPrior::default_(1))
                       .fit_kernel()
                       .fit_prior()
                       .train();
        const auto p = gp2.predict(input);
        std::printf("builder/exponential/linear prior: %.6f\n", p[0]);
        EXPECT(std::isfinite(p[0]), "builder prediction not finite");
    }
    // 2-D dataset (src/main.rs:57-68)
    {
        const std::vector<std::vector<double>> x2 = {{0.8, 0.1}, {1.2, 0.2}, {3.8, 0.3}, {4.2, 0.5}};
        auto gp3 = GaussianProcess<>::default_(x2, training_outputs);
        const std::vector<double> in2 = {1.0, 0.4};
        std::printf("2-D prediction: %.6f +- %.6f\n", gp3.predict(in2)[0], std::sqrt(gp3.predict_variance(in2)[0]));
    }
    // kernel arithmetic (kernel.rs:312-332) and the error texts of the reference
    {
        auto k = SquaredExp(0.8, 1.0) * Matern2(1.5, 0.5) + Linear(0.1);
        EXPECT(k.nb_parameters() == 5 && !k.is_scalable(), "composite kernel metadata");
        GaussianProcess<decltype(k), ZeroPrior> gp4(ZeroPrior{}, k, 0.1, false, 0.0, training_inputs, training_outputs);
        EXPECT(std::isfinite(gp4.predict(input)[0]), "composite kernel prediction");
        bool threw = false;
        try {
            GaussianProcess<HyperTan, ZeroPrior> bad(ZeroPrior{}, HyperTan(1.0, 0.0), 0.0, false, 0.0,
                                                    std::vector<std::vector<double>>{{1.0}, {2.0}, {3.0}}, {0.0, 0.0, 0.0});
        } catch (const std::runtime_error& e) {
            threw = std::string(e.what()).find("consider setting `cholesky_epsilon`") != std::string::npos;
        }
        EXPECT(threw, "indefinite kernel must raise the reference's panic text (algebra/mod.rs:90)");
        threw = false;
        try {
            gp.predict(std::vector<double>{1.0, 2.0});  // wrong width: mod.rs:231
        } catch (const std::runtime_error&) {
            threw = true;
        }
        EXPECT(threw, "shape assertion");
    }
    std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
