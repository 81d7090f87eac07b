// thfhe_keyslot.h -- lifetime + call combining for the per-key-set device contexts behind the libtfhe-named entry points.
//
// The reference's C++ programs call bootsXXX(result, a, b, cloud_key) from OpenMP threads on a shared key set
// (src/KNN_medical_data.cpp:681-691) and may delete a key set and load another one at the SAME address.  One Slot = the device
// context built from one key set + the queue that combines concurrent single-gate calls into one launch.  Rules:
//   * a Slot is held by std::shared_ptr for the whole of every call that uses it: neither a key swap at its address (fingerprint
//     mismatch -> the map entry is replaced) nor forget() can destroy the context, the mutex or the condition variable under a
//     caller; the context is destroyed by whoever lets go last;
//   * a replaced / forgotten Slot is `retired`: calls that already hold it finish on it (they were issued against that key),
//     new arrivals look the map up again and never see it.
// Header-only and independent of HIP so that tests/cpp/keyslot_test.cpp can drive it with a counting fake context under
// -fsanitize=thread (no GPU needed).
#pragma once

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace thfhe_slot {

template <class Ctx, class Fp, class Req>
struct Slot {
    Ctx *ctx = nullptr;
    void (*destroy)(Ctx *) = nullptr;
    Fp fp{};
    int n = 0;
    // combining queue
    std::mutex m;
    std::condition_variable cv;
    std::vector<Req *> queue;
    bool leader_active = false;
    Slot() = default;
    Slot(const Slot &) = delete;
    Slot &operator=(const Slot &) = delete;
    ~Slot() {
        if (ctx && destroy) destroy(ctx);
    }
};

template <class Ctx, class Fp, class Req>
class Cache {
  public:
    using SlotT = Slot<Ctx, Fp, Req>;
    using Ptr = std::shared_ptr<SlotT>;

    // The slot of the key set at address `key` whose fingerprint is `fp`; built by build(slot) (fills ctx / destroy / n; returns
    // false on failure) when there is none or when another key set now lives at that address.
    template <class Build>
    Ptr acquire(const void *key, const Fp &fp, Build &&build) {
        Ptr old;  // destroyed (if we hold the last reference) after the lock is released
        std::lock_guard<std::mutex> g(mu_);
        auto it = slots_.find(key);
        if (it != slots_.end()) {
            if (it->second->fp == fp) return it->second;
            old = std::move(it->second);  // callers that hold it keep a valid context; it dies with its last user
            slots_.erase(it);
        }
        Ptr s = std::make_shared<SlotT>();
        s->fp = fp;
        if (!build(*s)) return Ptr();
        slots_[key] = s;
        return s;
    }

    void forget(const void *key) {
        Ptr old;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = slots_.find(key);
            if (it == slots_.end()) return;
            old = std::move(it->second);
            slots_.erase(it);
        }
        // `old` goes out of scope here: the context is destroyed now if no call holds the slot, else by the last such call
    }

    size_t size() {
        std::lock_guard<std::mutex> g(mu_);
        return slots_.size();
    }

  private:
    std::mutex mu_;
    std::map<const void *, Ptr> slots_;
};

// One single-gate request: calls that arrive while a launch is in flight are queued and evaluated TOGETHER by the next leader
// (execute(slot, batch) runs without the queue lock), so T concurrent callers cost one gate latency, not T.  `req.done` must be
// a bool member; the caller keeps `slot` (a shared_ptr) alive across the call.
template <class SlotT, class Req, class Execute>
void combine(SlotT &slot, Req &req, Execute &&execute) {
    std::unique_lock<std::mutex> lk(slot.m);
    slot.queue.push_back(&req);
    while (!req.done) {
        if (!slot.leader_active) {
            slot.leader_active = true;
            std::vector<Req *> batch;
            batch.swap(slot.queue);
            lk.unlock();
            if (!batch.empty()) execute(slot, batch);
            lk.lock();
            for (Req *q : batch) q->done = true;
            slot.leader_active = false;
            slot.cv.notify_all();
        } else {
            slot.cv.wait(lk);
        }
    }
}

}  // namespace thfhe_slot
