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
    printf("ok\n");
    return 0;
}
