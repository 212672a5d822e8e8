// TEST INFRASTRUCTURE ONLY (oracle tool).  Prints the bucket-count growth chain of libstdc++ unordered_map
// (std::__detail::_Prime_rehash_policy) and cross-checks it against a live map.  Its output is the ORC_CHAIN /
// D3F_CHAIN table used by oracle/d3f_oracle.c and d3feat_amd/csrc/grid_subsample.hip.
// Build+run: g++ -O2 -o /tmp/chain oracle/tools/probe_libstdcxx_chain.cpp && /tmp/chain
#include <unordered_map>
#include <cstdio>
int main(){
  std::__detail::_Prime_rehash_policy pol;
  // growth chain as unordered_map follows it: 13, then next_bkt(2*nb)
  size_t nb = pol._M_next_bkt(12);
  printf("%zu", nb);
  while (nb < (size_t)1<<36) { nb = pol._M_next_bkt(2*nb); printf(", %zu", nb);} 
  printf("\n");
  // check by real map up to 2M
  std::unordered_map<size_t,int> m; size_t last=0;
  for (size_t i=0;i<3000000;i++){ m.emplace(i*7919u,0); if(m.bucket_count()!=last){ last=m.bucket_count(); printf("size %zu -> %zu\n", m.size(), last);} }
}
