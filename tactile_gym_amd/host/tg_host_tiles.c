/* tg_host_tiles.c - host side of the tile-sparse observation download (plain C + pthreads, no device code).
 *
 * The numpy VecEnv boundary (what the reference's sb3_helpers consume: stable-baselines3 VecEnv.step_wait() -> numpy observations,
 * sb3_helpers/rl_utils.py:17-30) copies the whole uint8 batch device -> host every step (16.8 MB for 1024 x 128 x 128).  With the tile
 * payload of csrc/tg_exchange.hip (tg_pack_tiles: only the 16 x 16 tiles that differ from the untouched sensor's image, 272-byte records
 * behind a 16-byte header) 1/10 of that crosses PCIe, and this function rebuilds the batch in a PERSISTENT host buffer the way
 * tg_unpack_tiles_multi does on rank 0's device: the tiles this buffer's previous frame had live get the template back, then the new
 * records land.  Message layout (parallel.py: TILE_MAGIC, TILE_REC): int32 {count, n_images, tiles_per_image, 0x54475431}, then count records
 * of {int32 tile id = image * tiles_per_image + tile, 12 bytes unused, 256 pixel bytes (16 rows x 16)}.
 *
 * Round 4: (1) every tile id - of the message and of the previous-frame list - is validated BEFORE the buffer is touched, so an error leaves
 * the buffer, the list and its count exactly as they were; (2) the rebuild runs on a small persistent thread pool (tg_host_pool_create):
 * thread t owns the images [t n / P, (t + 1) n / P) - it restores and scatters only tiles of its images, so no two threads write the same
 * byte - measured 0.126 ms -> see profiles/ for 1024 x 128 x 128 on one thread against four.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* spin-wait hint: x86 `pause`, AArch64 `yield`, a compiler barrier elsewhere (the bare x86 builtin did not compile on other hosts) */
#if defined(__x86_64__) || defined(__i386__)
#define TG_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define TG_CPU_RELAX() __asm__ __volatile__("yield" ::: "memory")
#else
#define TG_CPU_RELAX() __asm__ __volatile__("" ::: "memory")
#endif

#define TG_TILE_MAGIC 0x54475431
#define TG_TILE_REC 272
#define TG_MAX_THREADS 16

typedef struct {
    const uint8_t *msg, *tmpl; uint8_t* dst; const int32_t* prev_ids; int64_t n_prev, count; int32_t n, H, W, T, TW;
    int32_t first_thread;               /* 0: the caller takes share 0 and waits; 1: asynchronous, the workers alone share the images */
    int32_t by_record;                  /* 1: nothing to restore (n_prev == 0): the threads share the RECORDS - thread t scatters records
                                         * [t count / P, (t + 1) count / P) whatever image they belong to (a tile id occurs once per message, so no two threads
                                         * write the same byte) and writes their ids into new_ids - instead of every thread scanning all records for its images */
    int32_t* new_ids;
} job_t;

static void run_records(const job_t* j, int64_t k_lo, int64_t k_hi) {
    const size_t img_bytes = (size_t)j->H * j->W;
    const int32_t T = j->T, TW = j->TW, W = j->W;
    const uint8_t* rec = j->msg + 16 + (size_t)k_lo * TG_TILE_REC;
    for (int64_t k = k_lo; k < k_hi; ++k, rec += TG_TILE_REC) {
        int32_t id;
        memcpy(&id, rec, 4);
        const int32_t img = id / T, tile = id % T, ty = tile / TW, tx = tile % TW;
        uint8_t* d = j->dst + (size_t)img * img_bytes + (size_t)ty * 16 * W + (size_t)tx * 16;
        for (int r = 0; r < 16; ++r) memcpy(d + (size_t)r * W, rec + 16 + 16 * r, 16);
        j->new_ids[k] = id;
    }
}

static void run_range(const job_t* j, int32_t img_lo, int32_t img_hi) {
    const size_t img_bytes = (size_t)j->H * j->W;
    const int32_t T = j->T, TW = j->TW, W = j->W;
    for (int64_t k = 0; k < j->n_prev; ++k) {          /* the previous frame's live tiles: back to the untouched sensor's image */
        const int32_t id = j->prev_ids[k], img = id / T;
        if (img < img_lo || img >= img_hi) continue;
        const int32_t tile = id % T, ty = tile / TW, tx = tile % TW;
        const uint8_t* s = j->tmpl + (size_t)ty * 16 * W + (size_t)tx * 16;
        uint8_t* d = j->dst + (size_t)img * img_bytes + (size_t)ty * 16 * W + (size_t)tx * 16;
        for (int r = 0; r < 16; ++r) memcpy(d + (size_t)r * W, s + (size_t)r * W, 16);
    }
    const uint8_t* rec = j->msg + 16;
    for (int64_t k = 0; k < j->count; ++k, rec += TG_TILE_REC) {
        int32_t id;
        memcpy(&id, rec, 4);
        const int32_t img = id / T;
        if (img < img_lo || img >= img_hi) continue;
        const int32_t tile = id % T, ty = tile / TW, tx = tile % TW;
        uint8_t* d = j->dst + (size_t)img * img_bytes + (size_t)ty * 16 * W + (size_t)tx * 16;
        for (int r = 0; r < 16; ++r) memcpy(d + (size_t)r * W, rec + 16 + 16 * r, 16);
    }
}

/* ---- a persistent pool: workers sleep on a condition variable between calls (a short spin first: calls arrive every few hundred us) */
typedef struct tg_host_pool {
    int n_threads;                       /* workers + the caller */
    pthread_t th[TG_MAX_THREADS];
    pthread_mutex_t mu; pthread_cond_t cv_go, cv_done;
    volatile long generation; volatile int pending, stop;
    job_t job;
    int idx[TG_MAX_THREADS];
} tg_host_pool;
typedef struct { tg_host_pool* p; int t; } worker_arg;

static void share(const tg_host_pool* p, int t, int32_t* lo, int32_t* hi) {
    const int64_t n = p->job.n, f = p->job.first_thread, parts = p->n_threads - f;
    *lo = (int32_t)(n * (t - f) / parts); *hi = (int32_t)(n * (t - f + 1) / parts);
}
static void* worker(void* a_) {
    worker_arg* a = (worker_arg*)a_;
    tg_host_pool* p = a->p; const int t = a->t;
    free(a);
    long seen = 0;
    for (;;) {
        int spun = 0;
        while (p->generation == seen && !p->stop && spun < 4000) { if ((++spun & 63) == 0) sched_yield(); TG_CPU_RELAX(); }   /* yield: the caller may sit on this CPU */
        if (p->generation == seen && !p->stop) {
            pthread_mutex_lock(&p->mu);
            while (p->generation == seen && !p->stop) pthread_cond_wait(&p->cv_go, &p->mu);
            pthread_mutex_unlock(&p->mu);
        }
        if (p->stop) return NULL;
        seen = p->generation;
        __sync_synchronize();
        if (p->job.by_record) {
            const int64_t c = p->job.count, P = p->n_threads;
            run_records(&p->job, c * t / P, c * (t + 1) / P);
        } else {
            int32_t lo, hi; share(p, t, &lo, &hi);
            run_range(&p->job, lo, hi);
        }
        __sync_synchronize();
        if (__sync_sub_and_fetch(&p->pending, 1) == 0) { pthread_mutex_lock(&p->mu); pthread_cond_signal(&p->cv_done); pthread_mutex_unlock(&p->mu); }
    }
}
tg_host_pool* tg_host_pool_create(int32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > TG_MAX_THREADS) n_threads = TG_MAX_THREADS;
    tg_host_pool* p = (tg_host_pool*)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->n_threads = n_threads;
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_go, NULL); pthread_cond_init(&p->cv_done, NULL);
    for (int t = 1; t < n_threads; ++t) {
        worker_arg* a = (worker_arg*)malloc(sizeof *a);
        a->p = p; a->t = t;
        if (pthread_create(&p->th[t], NULL, worker, a) != 0) { free(a); p->n_threads = t; break; }
    }
    {   /* workers on the CPUs next to the creator's (same cache complex: where the host buffers were first touched).  Measured on the 256-CPU MI355X
         * host, 4 threads: 3.62 M env-steps/s pinned against 2.81 M with the workers wherever the scheduler puts them.  TG_HOST_PIN=0 leaves them free. */
        const char* pin_ = getenv("TG_HOST_PIN");
        if (!(pin_ && pin_[0] == '0')) {
        const int base = sched_getcpu();
        cpu_set_t allowed;
        if (base >= 0 && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
            for (int t = 1; t < p->n_threads; ++t) {
                const int cpu = (base & ~7) | ((base + t) & 7);
                if (!CPU_ISSET(cpu, &allowed)) continue;
                cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpu, &one);
                (void)pthread_setaffinity_np(p->th[t], sizeof one, &one);
            }
        }
        }
    }
    return p;
}
void tg_host_pool_destroy(tg_host_pool* p) {
    if (!p) return;
    pthread_mutex_lock(&p->mu); p->stop = 1; pthread_cond_broadcast(&p->cv_go); pthread_mutex_unlock(&p->mu);
    for (int t = 1; t < p->n_threads; ++t) pthread_join(p->th[t], NULL);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_go); pthread_cond_destroy(&p->cv_done);
    free(p);
}
int32_t tg_host_pool_threads(const tg_host_pool* p) { return p ? p->n_threads : 1; }

/* msg: the message (msg_bytes available); tmpl: uint8 [H*W]; dst: uint8 [n][H][W], holding what the previous call on it left;
 * prev_ids: int32 [n * (H/16) * (W/16)] capacity, *n_prev entries valid on entry (the tiles of dst that differ from tmpl), the new list on
 * return.  pool: NULL = this thread only.  Returns the record count, or -1 bad argument, -2 bad header, -3 message shorter than its count
 * says, -4 tile id out of range - and then dst, prev_ids and *n_prev are unchanged.  A tile id is assumed to occur at most ONCE per message
 * (what tg_pack_tiles produces: one record per live tile); this is not checked - a message that repeats an id makes two threads write the same
 * tile, whose final content is then one of the two records (memory safe, not deterministic). */
int64_t tg_host_unpack_tiles_mt(tg_host_pool* pool, const uint8_t* msg, int64_t msg_bytes, const uint8_t* tmpl, int32_t n, int32_t H, int32_t W,
                                uint8_t* dst, int32_t* prev_ids, int64_t* n_prev) {
    if (!msg || !tmpl || !dst || !prev_ids || !n_prev || n <= 0 || H <= 0 || W <= 0 || (H & 15) || (W & 15) || msg_bytes < 16) return -1;
    const int32_t TW = W / 16, T = (H / 16) * TW;
    int32_t hdr[4];
    memcpy(hdr, msg, 16);
    const int64_t count = hdr[0], total = (int64_t)n * T;
    if ((uint32_t)hdr[3] != (uint32_t)TG_TILE_MAGIC || hdr[1] != n || hdr[2] != T || count < 0 || count > total) return -2;
    if (16 + count * TG_TILE_REC > msg_bytes) return -3;
    if (*n_prev < 0 || *n_prev > total) return -1;
    for (int64_t k = 0; k < *n_prev; ++k) if (prev_ids[k] < 0 || prev_ids[k] >= total) return -4;      /* validate everything first */
    {
        const uint8_t* rec = msg + 16;
        for (int64_t k = 0; k < count; ++k, rec += TG_TILE_REC) { int32_t id; memcpy(&id, rec, 4); if (id < 0 || id >= total) return -4; }
    }
    job_t j = {msg, tmpl, dst, prev_ids, *n_prev, count, n, H, W, T, TW, 0, 0, prev_ids};
    if (pool && pool->n_threads > 1 && n >= pool->n_threads) {
        j.by_record = (*n_prev == 0);      /* the restore ran ahead (tg_host_restore_begin) or the buffer is fresh: only the scatter is left */
        pool->job = j;
        pool->pending = pool->n_threads - 1;
        __sync_synchronize();
        pthread_mutex_lock(&pool->mu); pool->generation++; pthread_cond_broadcast(&pool->cv_go); pthread_mutex_unlock(&pool->mu);
        if (j.by_record) run_records(&pool->job, 0, count / pool->n_threads);
        else { int32_t lo, hi; share(pool, 0, &lo, &hi); run_range(&pool->job, lo, hi); }
        int spun = 0;
        while (pool->pending > 0 && spun < 20000) { if ((++spun & 15) == 0) sched_yield(); TG_CPU_RELAX(); }   /* yield: a pinned worker may need this very CPU */
        if (pool->pending > 0) { pthread_mutex_lock(&pool->mu); while (pool->pending > 0) pthread_cond_wait(&pool->cv_done, &pool->mu); pthread_mutex_unlock(&pool->mu); }
        __sync_synchronize();
    } else {
        run_range(&j, 0, n);
    }
    if (!j.by_record || !(pool && pool->n_threads > 1 && n >= pool->n_threads)) {      /* (the record-sharing threads wrote the new list themselves) */
        const uint8_t* rec = msg + 16;
        for (int64_t k = 0; k < count; ++k, rec += TG_TILE_REC) memcpy(&prev_ids[k], rec, 4);
    }
    *n_prev = count;
    return count;
}
/* The restore half of the NEXT rebuild of a buffer, ahead of time and off the caller's thread: the tiles listed in prev_ids get the template back
 * on the pool's workers while the caller goes on (the device steps, Python runs); tg_host_pool_wait, then the buffer has no live tile (*n_prev
 * = 0 is the caller's to set) and the next tg_host_unpack_tiles_mt only scatters.  Without workers it runs here.  Returns 0, -1 bad argument,
 * -4 tile id out of range (nothing touched). */
int32_t tg_host_restore_begin(tg_host_pool* pool, const uint8_t* tmpl, int32_t n, int32_t H, int32_t W, uint8_t* dst, const int32_t* prev_ids, int64_t n_prev) {
    if (!tmpl || !dst || !prev_ids || n <= 0 || H <= 0 || W <= 0 || (H & 15) || (W & 15)) return -1;
    const int32_t TW = W / 16, T = (H / 16) * TW;
    const int64_t total = (int64_t)n * T;
    if (n_prev < 0 || n_prev > total) return -1;
    for (int64_t k = 0; k < n_prev; ++k) if (prev_ids[k] < 0 || prev_ids[k] >= total) return -4;
    static const uint8_t empty_msg[16] = {0};
    job_t j = {empty_msg, tmpl, dst, prev_ids, n_prev, 0, n, H, W, T, TW, 1, 0, NULL};
    if (pool && pool->n_threads > 1 && n >= pool->n_threads) {
        pool->job = j;
        pool->pending = pool->n_threads - 1;
        __sync_synchronize();
        pthread_mutex_lock(&pool->mu); pool->generation++; pthread_cond_broadcast(&pool->cv_go); pthread_mutex_unlock(&pool->mu);
    } else {
        j.first_thread = 0;
        run_range(&j, 0, n);
    }
    return 0;
}
void tg_host_pool_wait(tg_host_pool* pool) {
    if (!pool) return;
    int spun = 0;
    while (pool->pending > 0 && spun < 20000) { if ((++spun & 15) == 0) sched_yield(); TG_CPU_RELAX(); }   /* yield: a pinned worker may need this very CPU */
    if (pool->pending > 0) { pthread_mutex_lock(&pool->mu); while (pool->pending > 0) pthread_cond_wait(&pool->cv_done, &pool->mu); pthread_mutex_unlock(&pool->mu); }
    __sync_synchronize();
}

int64_t tg_host_unpack_tiles(const uint8_t* msg, int64_t msg_bytes, const uint8_t* tmpl, int32_t n, int32_t H, int32_t W, uint8_t* dst,
                             int32_t* prev_ids, int64_t* n_prev) {
    return tg_host_unpack_tiles_mt(NULL, msg, msg_bytes, tmpl, n, H, W, dst, prev_ids, n_prev);
}

/* uint8 [n][H][W] <- n copies of tmpl: the state a fresh buffer (no live tiles) must have. */
int32_t tg_host_fill_template(const uint8_t* tmpl, int32_t n, int32_t H, int32_t W, uint8_t* dst) {
    if (!tmpl || !dst || n <= 0 || H <= 0 || W <= 0) return -1;
    const size_t img_bytes = (size_t)H * W;
    for (int32_t i = 0; i < n; ++i) memcpy(dst + (size_t)i * img_bytes, tmpl, img_bytes);
    return 0;
}

/* checksum of a byte range (FNV-1a over 8-byte words): the read-only contract's debug check (vec_env.py: obs_guard) */
uint64_t tg_host_checksum(const uint8_t* p, int64_t nbytes) {
    uint64_t h = 1469598103934665603ull;
    int64_t i = 0;
    for (; i + 8 <= nbytes; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 1099511628211ull; }
    for (; i < nbytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
