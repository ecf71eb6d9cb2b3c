// Generates tests/golden/rocrand_philox_kat.json: known-answer vectors for the engine's RNG
// stream, produced by rocRAND 7.2's own Philox4x32-10 device API running on the HOST
// (the API is __host__ __device__).  rocRAND is the published third-party definition the
// hand-written Philox must reproduce; it is not part of the reference.
// Build+run (no GPU needed):  see tests/golden/make_golden.py
#include <cstdio>
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstdio>

int main()
{
    const unsigned long long seeds[] = {1234ULL, 1235ULL, 0x123456789abcdef0ULL};
    const unsigned long long subseqs[] = {0ULL, 1ULL, 99999ULL, 0x100000000ULL, 0xfedcba9876543210ULL};
    std::printf("{\n \"generator\": \"rocRAND %d Philox4x32-10, host execution\",\n \"cases\": [\n", ROCRAND_VERSION);
    bool first = true;
    for (unsigned long long seed : seeds)
        for (unsigned long long sub : subseqs) {
            // raw words of blocks 0..2
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, sub, 0, &st);
            unsigned int raw[12];
            for (int i = 0; i < 12; ++i) raw[i] = rocrand(&st);
            rocrand_init(seed, sub, 0, &st);
            float4 n4a = rocrand_normal4(&st);
            float4 n4b = rocrand_normal4(&st);
            rocrand_init(seed, sub, 0, &st);
            double2 d2a = rocrand_normal_double2(&st);
            double2 d2b = rocrand_normal_double2(&st);
            std::printf("%s  {\"seed\": \"%llu\", \"subsequence\": \"%llu\",\n   \"raw\": [", first ? "" : ",\n", seed, sub);
            for (int i = 0; i < 12; ++i) std::printf("%s%u", i ? ", " : "", raw[i]);
            std::printf("],\n   \"normal4\": [%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g],\n", n4a.x, n4a.y, n4a.z, n4a.w,
                        n4b.x, n4b.y, n4b.z, n4b.w);
            std::printf("   \"normal_double2\": [%.17g, %.17g, %.17g, %.17g]}", d2a.x, d2a.y, d2b.x, d2b.y);
            first = false;
        }
    std::printf("\n ]\n}\n");
    return 0;
}
