// Third-party items the reference imports on the DSP path, restated from their published sources so that the
// interpreter can run the reference's own code.  (std::num::Wrapping and the std methods are built into interp.py.)
//
// num-complex 0.4 (symphonia-core/Cargo.toml:40, re-exported as symphonia_core::dsp::complex, dsp/mod.rs:13):
// `Complex<T> { re, im }`, `new`, `conj`, `scale`, `norm_sqr` and the arithmetic operator impls of src/lib.rs --
// Add / Sub element-wise, Mul as (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re): two products and one sum per
// component, in that operand order, no fused multiply-add (the crate's `mul_add` paths are only used by explicit
// MulAdd calls, which the reference never makes).

pub struct Complex<T> {
    pub re: T,
    pub im: T,
}

impl<T> Complex<T> {
    pub fn new(re: T, im: T) -> Self {
        Complex { re, im }
    }

    pub fn conj(&self) -> Self {
        Complex { re: self.re, im: -self.im }
    }

    pub fn scale(&self, t: T) -> Self {
        Complex { re: self.re * t, im: self.im * t }
    }

    pub fn norm_sqr(&self) -> T {
        self.re * self.re + self.im * self.im
    }
}

impl<T> Default for Complex<T> {
    fn default() -> Self {
        Complex { re: 0.0, im: 0.0 }
    }
}

impl<T> Add for Complex<T> {
    fn add(self, other: Self) -> Self {
        Complex { re: self.re + other.re, im: self.im + other.im }
    }
}

impl<T> Sub for Complex<T> {
    fn sub(self, other: Self) -> Self {
        Complex { re: self.re - other.re, im: self.im - other.im }
    }
}

impl<T> Mul for Complex<T> {
    fn mul(self, other: Self) -> Self {
        let re = self.re * other.re - self.im * other.im;
        let im = self.re * other.im + self.im * other.re;
        Complex { re, im }
    }
}

impl<T> Neg for Complex<T> {
    fn neg(self) -> Self {
        Complex { re: -self.re, im: -self.im }
    }
}
