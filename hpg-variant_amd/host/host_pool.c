/* host_pool.c -- thread teams that sleep between jobs; NUMA placement of a run.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ------------------------------------------------------------------------ */
/* a small team of threads that sleep between jobs (condition variables, no spinning: a caller inside a CPU-limited      */
/* container must not burn its quota waiting): the runners' reader / formatter teams and a lone caller's staging        */
/* ------------------------------------------------------------------------ */


static void pool_drain(io_pool_t *p) {                  /* called with mu held; returns with mu held */
    while (p->next < p->n_tasks) {
        const int t = p->next++;
        pthread_mutex_unlock(&p->mu);
        p->fn(p->arg, t);
        pthread_mutex_lock(&p->mu);
    }
}

static void *pool_worker(void *v) {
    io_pool_t *p = (io_pool_t *)v;
    int seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->stop && p->gen == seen) pthread_cond_wait(&p->cv_work, &p->mu);
        if (p->stop) break;
        seen = p->gen;
        p->running++;
        pool_drain(p);
        if (--p->running == 0) pthread_cond_broadcast(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

/* n_threads counts the caller, which works too: n_threads - 1 threads are created */
void pool_init(io_pool_t *p, int n_threads) {
    memset(p, 0, sizeof *p);
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv_work, NULL);
    pthread_cond_init(&p->cv_done, NULL);
    if (n_threads > 1) p->th = (pthread_t *)calloc((size_t)n_threads - 1, sizeof(pthread_t));
    for (int i = 0; p->th && i < n_threads - 1; i++) {
        if (pthread_create(&p->th[p->n_threads], NULL, pool_worker, p) == 0) p->n_threads++;
    }
}

/* the pool's threads spread over the CPUs the process may use, one each (the caller's own CPU last): a thread woken from a
 * condition variable starts on its waker's CPU, and some schedulers (small VMs) leave it queued there for a whole slice
 * instead of moving it to an idle core -- a 2 ms job cannot wait for that */
void pool_spread(io_pool_t *p) {
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    const int self = sched_getcpu();
    int cpus[CPU_SETSIZE], n = 0;
    for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed) && c != self) cpus[n++] = c;
    if (self >= 0 && CPU_ISSET(self, &allowed)) cpus[n++] = self;
    if (n < 2) return;
    for (int i = 0; i < p->n_threads; i++) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cpus[i % n], &one);
        (void)pthread_setaffinity_np(p->th[i], sizeof one, &one);
    }
}

void pool_destroy(io_pool_t *p) {
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n_threads; i++) pthread_join(p->th[i], NULL);
    free(p->th);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_work); pthread_cond_destroy(&p->cv_done);
    memset(p, 0, sizeof *p);
}

/* runs fn(arg, 0 .. n_tasks-1), each task once, on the pool's threads and the caller; one job at a time per
 * pool.  p == NULL, or a pool without threads, runs the tasks inline. */
void pool_run(io_pool_t *p, pool_fn fn, void *arg, int n_tasks) {
    if (!p || p->n_threads == 0 || n_tasks <= 1) { for (int t = 0; t < n_tasks; t++) fn(arg, t); return; }
    pthread_mutex_lock(&p->mu);
    p->fn = fn; p->arg = arg; p->n_tasks = n_tasks; p->next = 0; p->gen++;
    pthread_cond_broadcast(&p->cv_work);
    p->running++;
    pool_drain(p);
    p->running--;
    while (p->running > 0) pthread_cond_wait(&p->cv_done, &p->mu);
    p->n_tasks = 0;
    pthread_mutex_unlock(&p->mu);
}

/* ------------------------------------------------------------------------ */
/* worker pools of the file-level code: persistent threads, tasks handed out  */
/* by a counter (a nested OpenMP team is created anew on every entry, which   */
/* costs milliseconds per batch on a many-core host)                           */
/* ------------------------------------------------------------------------ */

int default_io_threads(void) {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    FILE *q = fopen("/sys/fs/cgroup/cpu.max", "r");         /* a container may be allotted far fewer CPUs than it sees */
    if (q) {
        long quota = 0, period = 0;
        if (fscanf(q, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0 && quota / period < n) n = quota / period > 0 ? quota / period : 1;
        fclose(q);
    }
    int t = n >= 32 ? 16 : n >= 16 ? (int)n : n >= 4 ? (int)(n / 2) : 1;
    if (g_env.io_threads > 0) t = (int)g_env.io_threads;
    return t;
}

/* ---- NUMA: a run's threads and page-locked buffers go to the node the GPU hangs off ------------------------------
 * Measured on a two-socket MI355X host: 580 k variants/s with the process on the GPU's node, 334 k on the other one,
 * anything in between when left to the scheduler (the copy out of the page cache into the staging buffers and the
 * H2D DMA then cross the socket link).  The calling thread's affinity is narrowed to the node's CPUs for the run
 * (the threads it creates inherit it, its allocations are first touched there) and put back afterwards.
 * HPGV_NO_NUMA_BIND=1 switches this off. */
int numa_bind_to_device(cpu_set_t *saved) {
    if (g_env.no_numa_bind || !g_ctx) return 0;
    int node = -1;
    if (hpgv_device_numa_node(g_ctx, &node) != HPGV_OK || node < 0) return 0;
    char path[96], list[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const int ok = fgets(list, sizeof list, f) != NULL;
    fclose(f);
    if (!ok || sched_getaffinity(0, sizeof *saved, saved) != 0) return 0;
    cpu_set_t want;
    CPU_ZERO(&want);
    int n_set = 0;
    for (char *p = list; *p;) {                          /* "0-63,128-191" */
        char *e;
        long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET((int)c, saved)) { CPU_SET((int)c, &want); n_set++; }
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    if (n_set == 0) return 0;                            /* the caller is not allowed on that node: leave it alone */
    return sched_setaffinity(0, sizeof want, &want) == 0;
}
void numa_unbind(const cpu_set_t *saved, int bound) { if (bound) (void)sched_setaffinity(0, sizeof *saved, saved); }
