/* CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) for the TFRecord
 * framing read/written by calciumgan_amd/gan/utils/tfrecord.py -- the record
 * files of dataset/generate_tfrecords.py:146-153 (tf.io.TFRecordWriter) and
 * gan/utils/dataset_helper.py:147-182 (tf.data.TFRecordDataset).  Host code,
 * not on the GPU hot path.  Slicing-by-8 tables built on first use. */
#include <stddef.h>
#include <stdint.h>

static uint32_t T[8][256];
static int ready = 0;

static void init_tables(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    T[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int s = 1; s < 8; ++s) T[s][i] = (T[s - 1][i] >> 8) ^ T[0][T[s - 1][i] & 0xff];
  ready = 1;
}

/* crc32c of buf[0..n), continuing from `crc` (0 for a fresh checksum) */
uint32_t cg_crc32c(uint32_t crc, const unsigned char* buf, size_t n) {
  if (!ready) init_tables();
  uint32_t c = ~crc;
  while (n && ((uintptr_t)buf & 7)) {
    c = T[0][(c ^ *buf++) & 0xff] ^ (c >> 8);
    --n;
  }
  while (n >= 8) {
    const uint32_t lo = c ^ ((uint32_t)buf[0] | (uint32_t)buf[1] << 8 |
                             (uint32_t)buf[2] << 16 | (uint32_t)buf[3] << 24);
    c = T[7][lo & 0xff] ^ T[6][(lo >> 8) & 0xff] ^ T[5][(lo >> 16) & 0xff] ^
        T[4][lo >> 24] ^ T[3][buf[4]] ^ T[2][buf[5]] ^ T[1][buf[6]] ^ T[0][buf[7]];
    buf += 8;
    n -= 8;
  }
  while (n--) c = T[0][(c ^ *buf++) & 0xff] ^ (c >> 8);
  return ~c;
}
