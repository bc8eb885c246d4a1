#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ipcl/ipcl.hpp"
#include "detail.hpp"
using namespace ipcl;
template <class F> double best(F f, int reps=7){double b=1e30;for(int i=0;i<reps;++i){auto t0=std::chrono::steady_clock::now();f();auto t1=std::chrono::steady_clock::now();b=std::min(b,std::chrono::duration<double,std::micro>(t1-t0).count());}return b;}
int main(){
  const size_t N=8192;
  std::vector<BigNumber> v(N);
  for(size_t i=0;i<N;++i) v[i]=getRandomBN(4096);
  printf("budget %d\n", detail::max_host_threads());
  std::vector<uint64_t> flat; std::vector<BigNumber> u, c;
  for (const char* thr : {"1","8"}) {
    setenv("X","1",1);
    printf("threads_for: %d\n", detail::threads_for(N,512));
    printf("pack   %.0f us\n", best([&]{flat=detail::pack(v,64);}));
    printf("unpack %.0f us\n", best([&]{u=detail::unpack(flat,N,64);}));
    printf("copy   %.0f us\n", best([&]{c=detail::copy_texts(v);}));
    printf("stdcopy %.0f us\n", best([&]{std::vector<BigNumber> d(v); c.swap(d);}));
    printf("maxbits %.0f us\n", best([&]{volatile int b=detail::max_bits(v);(void)b;}));
    printf("PlainText(v).getTexts %.0f us\n", best([&]{ c = PlainText(v).getTexts(); }));
    break;
  }
}
