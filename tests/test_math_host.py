"""The branch-free fp64 activations of the kernels (csrc/cude_math.h) compiled for the host and compared with
libm.  (On the device the reciprocal seed is v_rcp_f64 instead of a float division; both are refined to
full precision by Newton steps.)"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "cude_math.h"
#include <cstdio>
#include <cmath>
#include <random>
int main(){
  std::mt19937_64 g(1); std::uniform_real_distribution<double> U(-1,1);
  double et=0, es=0, eg=0, ee=0;
  for(int i=0;i<2000000;i++){
    double s = std::pow(10.0, 3*U(g)-1.5); double x = U(g)*s*20;
    et=fmax(et,fabs(cude::m_tanh(x)-tanh(x)));
    double sg; double sp=cude::m_softplus(x,&sg);
    double ref = x>30? x + log1p(exp(-x)) : log1p(exp(x));
    es=fmax(es,fabs(sp-ref)/fmax(1.0,fabs(ref))); eg=fmax(eg,fabs(sg-1.0/(1.0+exp(-x))));
    double y=U(g)*40; ee=fmax(ee,fabs(cude::m_exp(y)/exp(y)-1));
  }
  double ev=0;
  for(int i=0;i<300000;i++){
    double z[6], t[6]; for(int j=0;j<6;j++){ double s = std::pow(10.0, 3*U(g)-1.5); z[j]=U(g)*s*20; }
    cude::m_tanh_vec<6>(z,t); for(int j=0;j<6;j++) ev=fmax(ev,fabs(t[j]-tanh(z[j])));
  }
  printf("%.6g %.6g %.6g %.6g %.6g\n",et,es,eg,ee,ev);
  double sg;
  printf("%.17g %.17g %.17g %.17g\n", cude::m_tanh(0.0), cude::m_tanh(900.0), cude::m_tanh(-1e9), cude::m_softplus(800.0,&sg));
}
'''


def test_activation_accuracy():
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-O2", "-I", os.path.join(ROOT, "conditional-ude_amd", "csrc"),
                               os.path.join(d, "t.cpp"), "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    et, es, eg, ee, ev = (float(v) for v in out[0].split())
    assert et < 5e-16 and es < 8e-16 and eg < 6e-16 and ee < 6e-16 and ev < 2e-15
    t0, tbig, tneg, spbig = (float(v) for v in out[1].split())
    assert t0 == 0.0 and tbig == 1.0 and tneg == -1.0 and spbig == 800.0
