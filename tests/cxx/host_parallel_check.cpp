// Test driver for fastq_utils_amd/host/fq_parallel.h (no GPU):
//   host_parallel_check gz <in> <out.gz> <level>     file -> multi-member gzip
//   host_parallel_check bgzf <in.bam> <out>          BGZF file -> inflated stream
//   host_parallel_check tobgzf <in> <out.bgzf> <split>   file, handed over as two pieces cut at <split>, -> BGZF file
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../fastq_utils_amd/host/fq_parallel.h"

static bool slurp(const char* path, std::vector<uint8_t>& v) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  uint8_t buf[1 << 16];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + k);
  fclose(f);
  return true;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::vector<uint8_t> in;
  if (!slurp(argv[2], in)) return 3;
  if (!strcmp(argv[1], "gz")) {
    fqhost::GzipMembers g;
    if (!g.open(argv[3], atoi(argv[4]))) return 4;
    // in two calls, as the programs write piece after piece
    const size_t half = in.size() / 2;
    if (!g.write((const char*)in.data(), half) || !g.write((const char*)in.data() + half, in.size() - half)) return 5;
    return g.close() ? 0 : 6;
  }
  std::vector<uint8_t> out;
  if (!strcmp(argv[1], "tobgzf")) {
    const size_t split = std::min<size_t>(in.size(), (size_t)atoll(argv[4]));
    if (!fqhost::bgzf_deflate_parallel({{in.data(), split}, {in.data() + split, in.size() - split}}, Z_DEFAULT_COMPRESSION, out)) return 9;
    FILE* f = fopen(argv[3], "wb");
    if (!f) return 8;
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return 0;
  }
  if (!fqhost::bgzf_inflate_parallel(in, out)) return 7;
  FILE* f = fopen(argv[3], "wb");
  if (!f) return 8;
  fwrite(out.data(), 1, out.size(), f);
  fclose(f);
  return 0;
}
