// keyslot_test.cpp -- CPU unit test of torus-fhe_amd/csrc/thfhe_keyslot.h (the context cache + call combiner behind bootsXXX),
// driven with a counting fake context instead of a HIP one; built with -fsanitize=thread by tests/test_keyslot.py.
//
// Scenario of the reference's callers: eight threads call single gates on a shared key set (src/KNN_medical_data.cpp:681-691)
// while a ninth keeps replacing the key set that lives at that address (delete + load at the same address) and sometimes forgets it.
// Checked: no call ever runs on a destroyed context, every call is evaluated under a key that lived at the address during the
// call, every context built is destroyed exactly once, calls are combined (fewer launches than calls).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../torus-fhe_amd/csrc/thfhe_keyslot.h"

struct FakeCtx {
    int key_version;
    std::atomic<int> alive{1};
    std::atomic<int> in_flight{0};
};
static std::atomic<int> g_built{0}, g_destroyed{0}, g_launches{0}, g_gates{0}, g_errors{0};

static void fake_destroy(FakeCtx *c) {
    if (c->in_flight.load() != 0) g_errors++;  // destroyed under a running launch
    if (c->alive.exchange(0) != 1) g_errors++;  // destroyed twice
    g_destroyed++;
    delete c;
}

struct Fp {
    int version;
    bool operator==(const Fp &o) const { return version == o.version; }
};
struct Req {
    int seen_version = -1;
    bool done = false;
};
using Cache = thfhe_slot::Cache<FakeCtx, Fp, Req>;

int main() {
    Cache cache;
    std::atomic<int> version{1};  // "the key set that currently lives at the address"
    std::atomic<bool> stop{false};
    const void *address = &version;
    const int kThreads = 8, kCalls = 400;

    auto build = [&](Cache::SlotT &s) {
        s.ctx = new FakeCtx;
        s.ctx->key_version = s.fp.version;
        s.destroy = fake_destroy;
        s.n = 1;
        g_built++;
        return true;
    };
    auto execute = [&](Cache::SlotT &s, const std::vector<Req *> &batch) {
        FakeCtx *c = s.ctx;
        c->in_flight++;
        if (c->alive.load() != 1) g_errors++;  // use after destroy
        std::this_thread::sleep_for(std::chrono::microseconds(200));  // a launch in flight: followers queue up behind it
        for (Req *r : batch) r->seen_version = c->key_version;
        if (c->alive.load() != 1) g_errors++;
        c->in_flight--;
        g_launches++;
        g_gates += (int)batch.size();
    };

    std::vector<std::thread> callers;
    for (int t = 0; t < kThreads; t++)
        callers.emplace_back([&] {
            for (int i = 0; i < kCalls; i++) {
                const int before = version.load();
                Cache::Ptr s = cache.acquire(address, Fp{before}, build);
                Req r;
                thfhe_slot::combine(*s, r, execute);
                const int after = version.load();
                if (!r.done || r.seen_version < before || r.seen_version > after) g_errors++;  // evaluated under a key of another era
            }
        });
    std::thread swapper([&] {
        while (!stop.load()) {
            std::this_thread::sleep_for(std::chrono::microseconds(700));
            const int v = version.fetch_add(1) + 1;
            if (v % 3 == 0) cache.forget(address);  // thfhe_tfhe_forget_key between loads
        }
    });
    for (auto &th : callers) th.join();
    stop = true;
    swapper.join();
    cache.forget(address);
    const int built = g_built.load(), destroyed = g_destroyed.load();
    std::printf("{\"built\": %d, \"destroyed\": %d, \"launches\": %d, \"gates\": %d, \"errors\": %d, \"slots_left\": %zu}\n", built, destroyed,
                g_launches.load(), g_gates.load(), g_errors.load(), cache.size());
    return (g_errors.load() == 0 && built == destroyed && built > 1 && g_gates.load() == kThreads * kCalls && cache.size() == 0) ? 0 : 1;
}
