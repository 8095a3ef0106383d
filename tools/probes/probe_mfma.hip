// Probe: validates MFMA fragment layouts and ds_read_b64_tr_b16 semantics on gfx950.
// Output is a text dump consumed by a human; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ inline unsigned short f2bf(float f){ unsigned u = __float_as_uint(f); u += 0x7fff + ((u>>16)&1); return (unsigned short)(u>>16); }

__global__ void k_mfma_bf16(const float* A, const float* B, float* C){
  // A: 16x32 row-major, B: 32x16 row-major, C: 16x16
  int l = threadIdx.x;
  bf16x8 a, b;
  for(int j=0;j<8;j++){
    a[j] = (short)f2bf(A[(l&15)*32 + (l>>4)*8 + j]);
    b[j] = (short)f2bf(B[((l>>4)*8 + j)*16 + (l&15)]);
  }
  f32x4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for(int r=0;r<4;r++) C[((l>>4)*4 + r)*16 + (l&15)] = c[r];
}
__global__ void k_mfma_f32(const float* A, const float* B, float* C){
  // A: 16x4, B: 4x16
  int l = threadIdx.x;
  float a = A[(l&15)*4 + (l>>4)];
  float b = B[(l>>4)*16 + (l&15)];
  f32x4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for(int r=0;r<4;r++) C[((l>>4)*4 + r)*16 + (l&15)] = c[r];
}
__global__ void k_tr(unsigned short* out, int mode){
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  int l = threadIdx.x;
  for(int i=l;i<4096;i+=64) lds[i] = (unsigned short)i;
  __syncthreads();
  int idx;
  if(mode==0) idx = l*4;                       // contiguous 8B per lane
  else { // hypothesised: 16-lane group g reads 4(k) x 16(n) block of a [k][64] row-major tile
    int g = l>>4, i = l&15;
    idx = (g*4 + (i>>2))*64 + (i&3)*4;          // row = g*4 + i/4, col = (i%4)*4
  }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + idx));
  for(int j=0;j<4;j++) out[l*4+j] = (unsigned short)v[j];
}
int main(){
  std::vector<float> A(16*32), B(32*16), C(256), R(256);
  for(int i=0;i<16;i++)for(int k=0;k<32;k++) A[i*32+k] = (float)((i*3+k*5)%17 - 8);
  for(int k=0;k<32;k++)for(int j=0;j<16;j++) B[k*16+j] = (float)((k*7+j*11)%13 - 6);
  float *dA,*dB,*dC; hipMalloc(&dA,4*512); hipMalloc(&dB,4*512); hipMalloc(&dC,4*256);
  hipMemcpy(dA,A.data(),4*512,hipMemcpyHostToDevice); hipMemcpy(dB,B.data(),4*512,hipMemcpyHostToDevice);
  k_mfma_bf16<<<1,64>>>(dA,dB,dC); hipMemcpy(C.data(),dC,4*256,hipMemcpyDeviceToHost);
  double err=0; for(int i=0;i<16;i++)for(int j=0;j<16;j++){ float s=0; for(int k=0;k<32;k++) s+=A[i*32+k]*B[k*16+j]; err=fmax(err,fabs(s-C[i*16+j])); }
  printf("mfma_bf16_16x16x32 maxerr %g\n", err);
  // f32
  std::vector<float> A4(64), B4(64);
  for(int i=0;i<16;i++)for(int k=0;k<4;k++) A4[i*4+k]=(float)((i*3+k*5)%17-8)+0.25f;
  for(int k=0;k<4;k++)for(int j=0;j<16;j++) B4[k*16+j]=(float)((k*7+j*11)%13-6)+0.5f;
  hipMemcpy(dA,A4.data(),4*64,hipMemcpyHostToDevice); hipMemcpy(dB,B4.data(),4*64,hipMemcpyHostToDevice);
  k_mfma_f32<<<1,64>>>(dA,dB,dC); hipMemcpy(C.data(),dC,4*256,hipMemcpyDeviceToHost);
  err=0; for(int i=0;i<16;i++)for(int j=0;j<16;j++){ float s=0; for(int k=0;k<4;k++) s+=A4[i*4+k]*B4[k*16+j]; err=fmax(err,fabs(s-C[i*16+j])); }
  printf("mfma_f32_16x16x4 maxerr %g\n", err);
  unsigned short* dO; hipMalloc(&dO, 2*256); std::vector<unsigned short> O(256);
  for(int mode=0;mode<2;mode++){
    k_tr<<<1,64>>>(dO,mode); hipMemcpy(O.data(),dO,512,hipMemcpyDeviceToHost);
    printf("tr mode %d:\n",mode);
    for(int l=0;l<64;l++){ printf(" l%02d: %4d %4d %4d %4d\n", l, O[l*4],O[l*4+1],O[l*4+2],O[l*4+3]); }
  }
  // simple HBM copy bandwidth
  size_t n = (size_t)1<<30; float *x,*y; hipMalloc(&x,n); hipMalloc(&y,n); hipMemset(x,1,n);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int it=0;it<3;it++){ hipEventRecord(e0); hipMemcpyAsync(y,x,n,hipMemcpyDeviceToDevice); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); printf("d2d copy 1GiB: %.3f ms -> %.1f GB/s (r+w)\n", ms, 2.0*n/ms/1e6); }
  return 0;
}
