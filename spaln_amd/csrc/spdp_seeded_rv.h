// spdp_seeded_rv.h -- where the walks of a seeded call (spdp_seeded.cpp, spdp_seeded_h.cpp) meet the thread that runs
// the device.
//
// A walk is deeply recursive host code that calls its DP engines synchronously from the inside (the reference runs one
// per worker thread, src/spaln.cc:1389-1468).  The device wants the opposite: many requests at once -- a launch lasts as
// long as its slowest problem, a few ms, whether it carries two hundred requests or two thousand.  So every walk runs
// on a FIBER (ucontext, own small stack): thousands are in flight on a handful of worker threads; a walk that reaches
// a DP call parks its request and switches back to its worker, which picks the next runnable walk.  A dispatcher (the
// calling thread, and a few more beside it, each with a lane context of its own) takes whatever is parked as soon as it
// is free (and either enough has gathered or no walk can run), runs it as one set of launches, and makes the owners
// runnable again -- host code of some walks overlaps the device batches of others, and several batches share the device.
#ifndef SPDP_SEEDED_RV_H_
#define SPDP_SEEDED_RV_H_
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <functional>
#include <memory>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "spdp_walk.h"
#include "spdp_hostcpus.h"

namespace spdp_seed {

struct Fiber;
struct Parked {
    int query = 0, kind = 0;
    Span s{}; SpdpWindow w{}; int cut[2] = {0, 0};
    int score = SPDP_NEVSEL;
    std::vector<SpdpSkl> rec;
    bool failed = false;
    int flags = 0;                              // SpdpAlignment::flags of the request
    Fiber* owner = nullptr;
    // a request handed over WITHOUT sleeping (Fiber::submit): nobody waits yet; `done` is set when it has been served, and a
    // walk that needs the result before that sleeps as `waiter` (Fiber::wait_for).  `async` and `waiter` under the scheduler's mutex;
    // `done` is also read by the walk without the lock (DeviceBackend::park): the dispatcher's store releases rec / score / flags,
    // the walk's load acquires them.
    bool async = false;
    struct Flag {
        std::atomic<bool> v{false};
        Flag() = default;
        Flag(const Flag& o) : v(o.v.load(std::memory_order_acquire)) {}
        Flag& operator=(const Flag& o) { v.store(o.v.load(std::memory_order_acquire), std::memory_order_release); return *this; }
        Flag& operator=(bool b) { v.store(b, std::memory_order_release); return *this; }
        operator bool() const { return v.load(std::memory_order_acquire); }
    } done;
    Fiber* waiter = nullptr;
};

struct Fiber {
    ucontext_t ctx{};
    ucontext_t* back = nullptr;                 // the worker it runs on right now
    void* stack = nullptr;
    int query = -1;
    int home = -1;                              // the worker thread this walk runs on, from its first step to its last
    Parked* want_park = nullptr;
    Parked* want_wait = nullptr;                // sleep until this submitted request has been served
    bool finished = false;
    struct WalkScheduler* sched = nullptr;
    // called by a walk (DpBackend::lsp / trcbk): hand the request over and sleep until it has been served
    void park(Parked* p)
    {
        p->owner = this;
        want_park = p;
        swapcontext(&ctx, back);                // (resumed by the same worker thread: see WalkScheduler::worker)
    }
    // (no switch: the requests wait in the fiber until it next goes back to its worker -- flush(), or any park / wait / end --
    // and are then taken over under one lock)
    std::vector<Parked*> to_submit;
    bool want_flush = false;
    void submit(Parked* p) { p->owner = nullptr; p->async = true; p->done = false; to_submit.push_back(p); }
    void flush() { if (!to_submit.empty()) { want_flush = true; swapcontext(&ctx, back); } }
    void wait_for(Parked* p) { want_wait = p; swapcontext(&ctx, back); }
};

// The requests of one walk by what defines them, for the scout pass (seeded_core, seeded_core_h): a walk is a deterministic function of its
// inputs and of the DP results it is given, so a second run asks for the same requests again.
struct RequestCache {
    struct Entry { int key[14]; Parked p; };
    std::vector<std::unique_ptr<Entry>> all;
    static void make_key(int* k, int kind, const Span& s, const SpdpWindow& w, const int* cut)
    {
        const int v[14] = {kind, s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr, w.lw, w.up, w.width, cut ? cut[0] : 0, cut ? cut[1] : 0};
        memcpy(k, v, sizeof v);
    }
    Entry* find(const int* k) { for (auto& e : all) if (!memcmp(e->key, k, sizeof e->key)) return e.get(); return nullptr; }
    Entry* add(const int* k) { all.emplace_back(new Entry); memcpy(all.back()->key, k, sizeof all.back()->key); return all.back().get(); }
    // ... and the HSP searches of the recursion levels by level and span
    struct Search { int key[9]; bool ok; std::vector<Unit> units; };
    std::vector<std::unique_ptr<Search>> searches;
    std::vector<int8_t> phs5, phs3;             // the phase marks the scout derived for the window (bind_problem)
};

// dispatcher lanes per latency class, shortest class first: "a,b,c,.." in SPDP_SEED_LANES overrides the defaults (a 0 merges
// that class into the one above it; "1" = one lane, no classes); a call of fewer than 64 walks gets one lane and one class.
// Returns the class each lane serves.
inline std::vector<int> lanes_per_class(int n_walks, std::vector<int> k)
{
    if (const char* e = getenv("SPDP_SEED_LANES")) {
        std::vector<int> v;
        for (const char* p = e; *p; ) { v.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        if (!v.empty()) k = v;
    }
    if (n_walks < 64) k.assign(1, 1);
    std::vector<int> class_of_lane;
    int cls = 0;
    for (size_t c = 0; c < k.size(); ++c) {
        const int m = std::max(c == 0 ? 1 : 0, std::min(k[c], 8));
        for (int j = 0; j < m; ++j) class_of_lane.push_back(cls);
        if (m) ++cls;
    }
    return class_of_lane;
}
// class of a request whose sweep takes `steps`: how many of the thresholds it reaches, within the classes that have lanes
inline int latency_class(int64_t steps, const int64_t* thr, int n_thr, int n_cls)
{
    static const std::vector<int64_t> env_thr = [] {            // SPDP_SEED_THR=a,b,..: other thresholds (tuning)
        std::vector<int64_t> v;
        if (const char* e = getenv("SPDP_SEED_THR"))
            for (const char* p = e; *p; ) { v.push_back(atoll(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        return v;
    }();
    if (!env_thr.empty()) { thr = env_thr.data(); n_thr = (int) env_thr.size(); }
    int c = 0;
    while (c < n_thr && steps >= thr[c]) ++c;
    return std::min(c, n_cls - 1);
}

struct WalkScheduler {
    static constexpr size_t STACK = 1 << 20, GUARD = 4096;     // (the HSP callback -- the reference's Wilip in an integration -- runs on it too)
    std::mutex mu;
    std::condition_variable cv_work, cv_main;
    // A walk never changes threads: compilers may keep thread-local addresses (errno, allocator caches, any thread_local
    // of the HSP callback -- the integrator's code runs on the fiber too) across a context switch, so a served walk goes
    // back to the queue of the worker that started it.
    std::vector<std::vector<Fiber*>> ready_of;          // per worker thread
    int n_ready = 0;
    std::vector<Fiber*> idle_fibers, all_fibers;
    std::vector<std::vector<Parked*>> parked;           // per latency class (= dispatcher lane)
    std::function<int(const Parked&)> classify;
    int n_walks = 0, next = 0, done = 0, in_flight = 0, busy = 0;
    int max_in_flight = 8192, n_threads = 16, batch_target = 256;
    bool all_in_flight = false;                         // every walk of the call gets its fiber at once (walks that hand their requests over and then wait)
    bool last_class_waits = false;                      // the dispatcher of the longest class starts only when every walk has handed its requests over
    std::atomic<int> scouted{0};                        // walks that have (Fiber::sched->scouted, counted by the walk body)
    std::function<void(int, Fiber&)> body;
    std::vector<int> order;                             // walk started k-th (empty: k); longest first shortens the tail of a call
    bool oom = false;
    int stack_limit = -1;                               // SPDP_SEED_TEST_STACKS: pretend mmap fails beyond so many fibers
    int64_t cpu_ns = 0;                                 // thread CPU time inside walks, all workers

    static void entry(unsigned lo, unsigned hi)
    {
        Fiber* f = (Fiber*) ((uintptr_t) lo | (uintptr_t) hi << 32);
        f->sched->body(f->query, *f);
        f->finished = true;
        swapcontext(&f->ctx, f->back);          // never resumed
    }
    Fiber* fresh(int q)
    {
        Fiber* f;
        if (!idle_fibers.empty()) { f = idle_fibers.back(); idle_fibers.pop_back(); }
        else {
            if (stack_limit >= 0 && (int) all_fibers.size() >= stack_limit) return nullptr;     // (test hook)
            void* m = mmap(nullptr, STACK + GUARD, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (m == MAP_FAILED) return nullptr;
            (void) mprotect(m, GUARD, PROT_NONE);       // a walk that outgrows its stack stops here, not in another walk's
            f = new Fiber;
            f->stack = m; f->sched = this;
            all_fibers.push_back(f);
        }
        f->query = q; f->finished = false; f->want_park = nullptr; f->want_wait = nullptr; f->want_flush = false; f->to_submit.clear();
        getcontext(&f->ctx);
        f->ctx.uc_stack.ss_sp = (char*) f->stack + GUARD;
        f->ctx.uc_stack.ss_size = STACK;
        f->ctx.uc_link = nullptr;
        const uintptr_t p = (uintptr_t) f;
        makecontext(&f->ctx, (void (*)()) entry, 2, (unsigned) (p & 0xffffffffu), (unsigned) (p >> 32));
        return f;
    }
    void worker(int me)
    {
        ucontext_t here;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            Fiber* f = nullptr;
            for (;;) {
                if (!ready_of[me].empty()) { f = ready_of[me].back(); ready_of[me].pop_back(); --n_ready; break; }
                if (next < n_walks && in_flight < max_in_flight && !oom) {
                    f = fresh(order.empty() ? next : order[next]);
                    if (f) { f->home = me; ++next; ++in_flight; break; }
                    // no stack for another walk (address space, vm.max_map_count): those in flight are what there is, the
                    // rest start as their fibers come free; not even one: the call fails
                    if (in_flight > 0) max_in_flight = in_flight;
                    else { oom = true; done = n_walks; next = n_walks; cv_main.notify_all(); cv_work.notify_all(); return; }
                }
                if (done >= n_walks) return;
                cv_work.wait(lk);
            }
            ++busy;
            lk.unlock();
            f->back = &here;
            timespec t0, t1;
            clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0);
            swapcontext(&here, &f->ctx);
            clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1);
            lk.lock();
            cpu_ns += (int64_t) (t1.tv_sec - t0.tv_sec) * 1000000000 + (t1.tv_nsec - t0.tv_nsec);
            --busy;
            if (!f->to_submit.empty()) {                // what the walk handed over without sleeping
                bool enough = false;
                for (Parked* p : f->to_submit) {
                    const int c = parked.size() > 1 ? std::max(0, std::min((int) parked.size() - 1, classify(*p))) : 0;
                    parked[c].push_back(p);
                    enough = enough || (int) parked[c].size() >= batch_target;
                }
                f->to_submit.clear();
                if (enough) cv_main.notify_all();
            }
            if (f->finished) {
                idle_fibers.push_back(f);
                --in_flight; ++done;
                if (done >= n_walks) { cv_work.notify_all(); cv_main.notify_all(); }
                else if (next < n_walks) cv_work.notify_one();
            } else if (f->want_flush) {                 // only came to hand its requests over
                f->want_flush = false;
                ready_of[me].push_back(f); ++n_ready;
            } else if (f->want_wait) {                  // needs a request it handed over earlier
                Parked* p = f->want_wait;
                f->want_wait = nullptr;
                if (p->done) { ready_of[me].push_back(f); ++n_ready; }
                else p->waiter = f;
            } else {
                const int c = parked.size() > 1 ? std::max(0, std::min((int) parked.size() - 1, classify(*f->want_park))) : 0;
                parked[c].push_back(f->want_park);
                if ((int) parked[c].size() >= batch_target) cv_main.notify_all();
            }
            if (!busy) cv_main.notify_all();
        }
    }
    // runs walks 0 .. n - 1 (body(q, fiber) on a fiber each).  A batch lasts as long as its slowest request, so requests
    // are sorted into latency classes (cls(request), short to long) and every class has dispatchers of its own -- lane l
    // serves class_of_lane[l]; lane 0 is this thread, the others are threads beside it: device(batch, lane) is called with the
    // parked requests of one class and fills score / rec / failed of each; a walk whose requests are short does not wait for
    // another walk's 20 kb intron.  Returns false when not every walk could be started.
    bool run(int n, std::function<void(int, Fiber&)> walk_body,
             const std::function<void(std::vector<Parked*>&, int)>& device, const std::vector<int>& class_of_lane,
             std::function<int(const Parked&)> cls)
    {
        const int n_lanes = (int) class_of_lane.size();
        parked.assign(*std::max_element(class_of_lane.begin(), class_of_lane.end()) + 1, std::vector<Parked*>());
        classify = std::move(cls);
        n_walks = n; body = std::move(walk_body);
        // walks in flight / mean latency of a walk = the rate of a long call: more in flight for big calls (each fiber is two
        // mappings: stay well below vm.max_map_count), and batches in proportion (60 000 pairs: 8192 / 256 -> 0.97 s, 32768 / 1024 -> 0.76 s)
        max_in_flight = all_in_flight ? std::min(28000, std::max(8192, n)) : std::min(20000, std::max(8192, n / 2));
        if (const char* e = getenv("SPDP_SEED_WALKS")) max_in_flight = std::max(1, atoi(e));
        batch_target = std::max(256, max_in_flight / 32);
        if (const char* e = getenv("SPDP_SEED_BATCH")) batch_target = std::max(1, atoi(e));
        n_threads = std::min(32, spdp_host_cpus());     // (threads beyond the CPUs granted only contend: 16 granted, measured 8 / 16 / 32 / 64)
        if (const char* e = getenv("SPDP_SEED_THREADS")) n_threads = std::max(1, atoi(e));
        if (const char* e = getenv("SPDP_SEED_TEST_STACKS")) stack_limit = atoi(e);
        n_threads = std::min(n_threads, n);
        std::vector<std::thread> pool;
        ready_of.assign(n_threads, std::vector<Fiber*>());
        n_ready = 0;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back([this, t] { worker(t); });
        auto dispatcher = [&](int lane) {
            const int c = class_of_lane[lane];
            for (;;) {
                std::vector<Parked*> take;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_main.wait(lk, [&] {
                        if (done >= n_walks) return true;
                        if (parked[c].empty()) return false;
                        // nobody can run (every worker waits: nothing ready, nothing new to start), or enough has gathered
                        const bool startable = next < n_walks && in_flight < max_in_flight && !oom;
                        const bool idle = !busy && n_ready == 0 && !startable;
                        // (the longest class while walks still hand requests over: its launches are long whatever their size, so few
                        // and full ones -- but from a thousand requests on the device is busy with them anyway)
                        if (last_class_waits && parked.size() > 1 && c == (int) parked.size() - 1)
                            return idle || scouted.load() >= n_walks || (int) parked[c].size() >= std::max(1024, batch_target);
                        return (int) parked[c].size() >= batch_target || idle;
                    });
                    if (parked[c].empty()) { cv_main.notify_all(); break; }         // every walk has ended
                    take.swap(parked[c]);
                }
                device(take, lane);
                {
                    std::lock_guard<std::mutex> g(mu);
                    for (Parked* p : take) {
                        if (p->async) {
                            p->done = true;
                            if (p->waiter) { ready_of[p->waiter->home].push_back(p->waiter); ++n_ready; p->waiter = nullptr; }
                        } else { ready_of[p->owner->home].push_back(p->owner); ++n_ready; }
                    }
                }
                cv_work.notify_all();
            }
        };
        std::vector<std::thread> lanes;
        for (int l = 1; l < n_lanes; ++l) lanes.emplace_back(dispatcher, l);
        dispatcher(0);
        for (std::thread& t : lanes) t.join();
        for (std::thread& t : pool) t.join();
        for (Fiber* f : all_fibers) { munmap(f->stack, STACK + GUARD); delete f; }
        all_fibers.clear();
        return !oom;
    }
};

}   // namespace spdp_seed
#endif
