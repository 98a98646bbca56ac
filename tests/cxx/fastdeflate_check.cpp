// TEST DRIVER: the fast deflate compressor (fastq_utils_amd/host/fq_fastdeflate.h) must write what zlib inflates back to
// the input.  argv: file member_bytes.  The file is cut into members of member_bytes (as GzipMembers cuts its text),
// every member goes through gzip_member_fast, the concatenation through zlib's inflate; exit 0 = the same bytes.
// stdout: "ok|differ in=.. out=.. members=.."
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../fastq_utils_amd/host/fq_fastdeflate.h"

int main(int argc, char** argv) {
  if (argc < 3) return 9;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 8;
  fseek(f, 0, SEEK_END);
  const size_t n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> d(n + 1);
  if (n && fread(d.data(), 1, n, f) != n) return 7;
  fclose(f);
  const size_t member = strtoull(argv[2], nullptr, 10);
  std::vector<uint8_t> all, m;
  size_t members = 0;
  for (size_t o = 0; o < n || members == 0; o += member) {
    const size_t len = n > o ? (n - o < member ? n - o : member) : 0;
    fqhost::fdef::gzip_member_fast(d.data() + o, len, m);
    all.insert(all.end(), m.begin(), m.end());
    ++members;
    if (!len) break;
  }
  std::vector<char> back;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  inflateInit2(&zs, 15 + 16);
  zs.next_in = all.data();
  zs.avail_in = (uInt)all.size();
  std::vector<char> buf(1 << 20);
  for (;;) {
    zs.next_out = (Bytef*)buf.data();
    zs.avail_out = (uInt)buf.size();
    const int rc = inflate(&zs, Z_NO_FLUSH);
    back.insert(back.end(), buf.data(), buf.data() + (buf.size() - zs.avail_out));
    if (rc == Z_STREAM_END) {
      if (zs.avail_in == 0) break;
      inflateReset(&zs);
      continue;
    }
    if (rc != Z_OK) {
      printf("differ inflate error %d %s\n", rc, zs.msg ? zs.msg : "");
      return 1;
    }
  }
  inflateEnd(&zs);
  const bool same = back.size() == n && (n == 0 || memcmp(back.data(), d.data(), n) == 0);
  printf("%s in=%zu out=%zu members=%zu\n", same ? "ok" : "differ", n, all.size(), members);
  return same ? 0 : 1;
}
