/*
 * bitstages.hip -- the reference's bit-level stages (algorithms/crc16.c, interleave.c, bit-scramble.c),
 * batched over independent packets.  Integer/byte work, bit-exact by construction;
 * nothing here is reshaped into a matrix product: it is a few bytes per packet.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace qpsk {

/* crc16.c:11-23, one lane per packet.  Packets are a few tens of bytes (the interleaver's prime table stops at
 * 347 bits), so the serial byte loop is short; consecutive lanes read consecutive packets. */
__global__ void __launch_bounds__(256)
crc16_kernel(const uint8_t *__restrict__ data, int npackets, int nbytes, uint16_t *__restrict__ crc_out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npackets) return;
    const uint8_t *d = data + (size_t)p * nbytes;
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < nbytes; i++) {
        uint8_t x = (uint8_t)((crc >> 8) ^ d[i]);
        x ^= (uint8_t)(x >> 4);
        crc = (uint16_t)((crc << 8) ^ ((uint16_t)(x << 12)) ^ ((uint16_t)(x << 5)) ^ (uint16_t)x);
    }
    crc_out[p] = crc;
}

/* interleave.c:49-75, one workgroup per packet, one lane per bit.  out starts at zero and receives OR-ed bits
 * (interleave.c:37,73): when b does not divide... when gcd(b, nbits) > 1 several source bits land on one
 * destination and are OR-ed, exactly as in the reference, so the scatter uses atomicOr on an LDS image. */
__global__ void __launch_bounds__(256)
interleave_kernel(uint8_t *data, int nbytes, unsigned b, int dir)
{
    extern __shared__ unsigned img[];                       /* ceil(nbytes / 4) words */
    uint8_t *pkt = data + (size_t)blockIdx.x * nbytes;
    const unsigned nbits = (unsigned)nbytes * 8u;
    const int nwords = (nbytes + 3) / 4;
    for (int w = threadIdx.x; w < nwords; w += blockDim.x) img[w] = 0;
    __syncthreads();
    for (unsigned n = threadIdx.x; n < nbits; n += blockDim.x) {
        unsigned i = n, j = (b * n) % nbits;
        if (dir == 1) { const unsigned t = j; j = i; i = t; }
        const unsigned bit = (pkt[i >> 3] >> (i & 7)) & 1u;
        if (bit) atomicOr(&img[j >> 5], 1u << (j & 31));   /* little-endian: bit j of byte j/8 == bit j%32 of word j/32 */
    }
    __syncthreads();
    const uint8_t *src = reinterpret_cast<const uint8_t *>(img);
    for (int k = threadIdx.x; k < nbytes; k += blockDim.x) pkt[k] = src[k];
}

/* bit-scramble.c:57-69: the scrambler is additive (the register is fed by its own taps, not by the data), so
 * a frame's keystream is a fixed 2-bit-per-symbol table built on the host from SEED; the kernel is one xor. */
__global__ void __launch_bounds__(256)
scramble_kernel(uint8_t *sym, const uint8_t *__restrict__ keystream, int npackets, int nsym)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)npackets * nsym) return;
    const uint8_t k = keystream[i % nsym];
    /* scramble_internal() rewrites bits 0 and 1 only and leaves the others (bit-scramble.c:62-63) */
    sym[i] = (uint8_t)(sym[i] ^ k);
}

/* Four 2-bit symbols per byte (round 6): a receive step's symbols are 1 byte each by the API (the value of qpsk_demod()'s two bits,
 * qpsk.c:77-78), but carry 2 bits -- and the copy-back over PCIe, not the kernel, bounds a host that gathers every step.
 * Row r of nsym symbols -> row r of ceil(nsym / 4) bytes: byte k = sym[4k] | sym[4k+1] << 2 | sym[4k+2] << 4 | sym[4k+3] << 6. */
__global__ void __launch_bounds__(256)
pack_dibits_kernel(const uint8_t *__restrict__ sym, uint8_t *__restrict__ packed, size_t nrows, int nsym)
{
    const int pb = (nsym + 3) / 4;
    if ((nsym & 15) == 0) {      /* 16 symbols in, 4 bytes out per thread */
        const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, per_row = (size_t)(nsym >> 4);
        if (i >= nrows * per_row) return;
        const uint4 v = reinterpret_cast<const uint4 *>(sym)[i];
        auto pk = [](unsigned w) { return (w & 3u) | ((w >> 6) & 0xcu) | ((w >> 12) & 0x30u) | ((w >> 18) & 0xc0u); };
        reinterpret_cast<uint32_t *>(packed)[i] = pk(v.x) | (pk(v.y) << 8) | (pk(v.z) << 16) | (pk(v.w) << 24);
        return;
    }
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * (size_t)pb) return;
    const size_t r = i / (size_t)pb;
    const int k = (int)(i - r * (size_t)pb);
    const uint8_t *row = sym + r * (size_t)nsym;
    unsigned b = 0;
    for (int j = 0; j < 4; j++)
        if (4 * k + j < nsym) b |= (unsigned)(row[4 * k + j] & 3u) << (2 * j);
    packed[i] = (uint8_t)b;
}

int launch_pack_dibits(const uint8_t *sym, uint8_t *packed, size_t nrows, int nsym, hipStream_t s)
{
    const size_t n = (nsym & 15) == 0 ? nrows * (size_t)(nsym >> 4) : nrows * (size_t)((nsym + 3) / 4);
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_dibits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sym, packed, nrows, nsym);
    return (int)hipGetLastError();
}

int launch_crc16(const uint8_t *data, int npackets, int nbytes, uint16_t *crc, hipStream_t s)
{
    hipLaunchKernelGGL(crc16_kernel, dim3((npackets + 255) / 256), dim3(256), 0, s, data, npackets, nbytes, crc);
    return (int)hipGetLastError();
}

int launch_interleave(uint8_t *data, int npackets, int nbytes, unsigned b, int dir, hipStream_t s)
{
    hipLaunchKernelGGL(interleave_kernel, dim3(npackets), dim3(256), sizeof(unsigned) * (size_t)((nbytes + 3) / 4), s, data,
                       nbytes, b, dir);
    return (int)hipGetLastError();
}

int launch_scramble(uint8_t *sym, const uint8_t *keystream, int npackets, int nsym, hipStream_t s)
{
    const size_t n = (size_t)npackets * nsym;
    hipLaunchKernelGGL(scramble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sym, keystream, npackets, nsym);
    return (int)hipGetLastError();
}

} // namespace qpsk
