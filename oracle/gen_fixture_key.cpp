// TEST INFRASTRUCTURE.  Regenerates the LWE secret key behind the reference's committed
// ciphertext fixtures (test/bootstrap_modules/*.data): the reference seeds libtfhe's generator
// with {100, 20032, 21341} (src/bootstrap_modules.cpp:52-55, src/libthfhe.cpp:362-363) and the LWE
// key is the first n=630 draws of uniform_int_distribution<int32_t>(0,1) on std::default_random_engine
// seeded from a std::seed_seq (libtfhe lwekey generation; verified: all 11 fixture files decrypt).
// Output: 630 characters '0'/'1' on one line (committed as tests/golden/fixture_lwe_key.txt).
#include <cstdint>
#include <cstdio>
#include <random>
int main() {
    uint32_t seed[] = {100, 20032, 21341};
    std::seed_seq q(seed, seed + 3);
    std::default_random_engine g;
    g.seed(q);
    std::uniform_int_distribution<int32_t> d(0, 1);
    for (int i = 0; i < 630; i++) putchar('0' + d(g));
    putchar('\n');
    return 0;
}
