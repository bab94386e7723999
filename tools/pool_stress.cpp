// CPU-only stress of the copy pool of ws_capi.cpp (the class is pasted below by tools/pool_stress.sh):
//   g++ -O2 -std=c++17 -pthread -fsanitize=thread pool_stress_gen.cpp && ./a.out
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <pthread.h>
#include <unistd.h>
#include <sys/wait.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
//@POOL@
int main()
{
    const size_t N = 9 << 20;
    std::vector<uint8_t> a(N), b(N);
    for (int it = 0; it < 400; ++it) {
        for (size_t i = 0; i < N; i += 4099) a[i] = (uint8_t)(it + i);
        size_t n = (size_t)(300 << 10) + (rand() % (8 << 20));
        memset(b.data(), 0, n);
        CopyPool::get().copy(b.data(), a.data(), n);
        if (memcmp(a.data(), b.data(), n)) { printf("MISMATCH at iteration %d\n", it); return 1; }
    }
    std::vector<uint8_t> c(N), d(N); // two threads submitting at once
    std::thread t([&] { for (int it = 0; it < 200; ++it) CopyPool::get().copy(d.data(), c.data(), N); });
    for (int it = 0; it < 200; ++it) {
        CopyPool::get().copy(b.data(), a.data(), N);
        if (memcmp(a.data(), b.data(), N)) { printf("MISMATCH2\n"); return 1; }
    }
    t.join();
    // the widening copies (the wire formats of ws_search_host): int16 -> double / float, float -> double
    {
        const size_t M = (3 << 20) + 77;
        std::vector<int16_t> w(M);
        std::vector<double> d64(M + 3);
        std::vector<float> f32(M + 3);
        for (size_t i = 0; i < M; ++i) w[i] = (int16_t)((i * 2654435761u) >> 16);
        for (int it = 0; it < 20; ++it) {
            CopyPool::get().copy(reinterpret_cast<uint8_t *>(d64.data() + (it & 3)), reinterpret_cast<const uint8_t *>(w.data()), M - (it & 3), kCopyI16F64);
            CopyPool::get().copy(reinterpret_cast<uint8_t *>(f32.data() + (it & 1)), reinterpret_cast<const uint8_t *>(w.data()), M - 8, kCopyI16F32);
            for (size_t i = 0; i < M - 8; i += 1 + i / 64) {
                if (d64[i + (it & 3)] != (double)w[i] || f32[i + (it & 1)] != (float)w[i]) { printf("WIDEN MISMATCH at %zu\n", i); return 1; }
            }
            CopyPool::get().copy(reinterpret_cast<uint8_t *>(d64.data()), reinterpret_cast<const uint8_t *>(f32.data()), M, kCopyF32F64);
            for (size_t i = 0; i < M; i += 997) if (d64[i] != (double)f32[i]) { printf("F32->F64 MISMATCH\n"); return 1; }
        }
    }
    // a forked child has the pool object but none of its threads: it must copy alone, not wait for helpers
    {
        std::thread busy([&] { for (int it = 0; it < 50; ++it) CopyPool::get().copy(d.data(), c.data(), N); });
        const pid_t pid = fork();
        if (pid == 0) {
            alarm(20); // (a child that waits for the parent's helpers would hang: the alarm ends it with a signal)
            CopyPool::get().copy(b.data(), a.data(), N);
            _exit(memcmp(a.data(), b.data(), N) ? 3 : 0);
        }
        int status = 0;
        waitpid(pid, &status, 0);
        busy.join();
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { printf("FORKED CHILD FAILED (status %d)\n", status); return 1; }
    }
    printf("ok\n");
    return 0;
}
