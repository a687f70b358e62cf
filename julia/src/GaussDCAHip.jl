# GaussDCAHip.jl -- the reference-side binding of libgdca.so.
#
# Drop-in for the hot path of GaussDCA.jl (src/GaussDCA.jl:28-42): same exported names, same
# keyword arguments and defaults as `gDCA` (src/GaussDCA.jl:8-16); the DCAUtils-named operators
# used by the reference (compute_weights, compute_weighted_frequencies, add_pseudocount,
# compute_FN, compute_DI_gauss) are thin `ccall` wrappers over include/gdca.h.
#
# The host pieces of the reference that are NOT on the hot path are not restated here: argument checks, ranking and
# printing are the reference's own functions (GaussDCA.check_arguments / compute_ranking / printrank,
# src/GaussDCA.jl:49-65, :88-99, :67-74), FASTA reading and duplicate removal are DCAUtils'.
#
# NOT EXECUTED in the build image (no Julia toolchain there); kept minimal and literal.  The struct layouts and
# status codes below are checked mechanically against include/gdca.h by
# tests/test_cabi_cpu.py::test_struct_layouts_agree_across_c_ctypes_and_julia.
module GaussDCAHip

export gDCA, gDCA_stepwise, printrank, compute_weights, compute_weighted_frequencies, add_pseudocount,
       compute_FN, compute_DI_gauss

using LinearAlgebra
import DCAUtils                                   # host-side I/O only: read_fasta_alignment, remove_duplicate_sequences
import GaussDCA: check_arguments, compute_ranking, printrank   # unchanged host code of the reference

const libgdca = get(ENV, "LIBGDCA", "libgdca.so")

struct GdcaParams
    pseudocount::Cdouble
    theta::Cdouble      # < 0  =>  :auto
    score::Int32        # 0 = :frob, 1 = :DI
    apc::Int32
end

mutable struct GdcaStats
    theta::Cdouble; Meff::Cdouble; pair_identity_sum::UInt64
    thresh::Int32; info::Int32; N::Int32; M::Int32; q::Int32; n::Int32; n_pad::Int32; update_launches::Int32
    inverse_batch::Int32; refined::Int32
    ms_total::Cdouble; ms_theta::Cdouble; ms_weights::Cdouble; ms_covariance::Cdouble
    ms_inverse::Cdouble; ms_inverse_update::Cdouble; ms_score::Cdouble
    inverse_flops::Cdouble; update_flops::Cdouble
    sweep_ghz::Cdouble; inverse_norm1::Cdouble; matrix_norm1::Cdouble; cond_bound::Cdouble
    ms_fn::Cdouble; ms_pair_tally::Cdouble
    sweep_retries::Int32; reserved0::Int32
    GdcaStats() = new()
end

const CTX = Ref{Ptr{Cvoid}}(C_NULL)

function ctx()
    if CTX[] == C_NULL
        dev = parse(Int32, get(ENV, "GDCA_DEVICE", "0"))
        st = ccall((:gdca_ctx_create, libgdca), Cint, (Int32, Ref{Ptr{Cvoid}}), dev, CTX)
        st == 0 || error("gdca_ctx_create failed (status $st): no usable HIP device; there is no CPU fallback")
        # the library writes ITS sizeof(gdca_stats) into our struct: refuse a build whose layout is not the one mirrored above
        ver = ccall((:gdca_version, libgdca), Int32, ())
        sb = ccall((:gdca_stats_bytes, libgdca), Int32, ())
        pb = ccall((:gdca_params_bytes, libgdca), Int32, ())
        (ver == 6 && sb == sizeof(GdcaStats) && pb == sizeof(GdcaParams)) ||
            error("libgdca is version $ver with gdca_stats of $sb bytes; this binding mirrors version 6 with $(sizeof(GdcaStats)) bytes")
    end
    return CTX[]
end

function check(st::Integer, info::Integer = 0)
    st == 0 && return
    msg = unsafe_string(ccall((:gdca_last_error, libgdca), Cstring, (Ptr{Cvoid},), ctx()))
    st == 1 && throw(ArgumentError(msg))
    st == 2 && throw(PosDefException(info))
    st == 4 && throw(OutOfMemoryError())
    st == 5 && throw(LinearAlgebra.LAPACKException(info))   # what eigvals() raises inside compute_DI_gauss (:37)
    error("libgdca: $msg")
end

theta_arg(θ) = θ === :auto ? -1.0 : Float64(θ)

# tuning switch of the context (include/gdca.h: gdca_ctx_set_option), e.g. set_option("REFINE", "0") or set_option("GROUP", 4);
# the GDCA_* environment variables are read once, when the context is created
function set_option(key::AbstractString, value)
    check(ccall((:gdca_ctx_set_option, libgdca), Cint, (Ptr{Cvoid}, Cstring, Cstring), ctx(), String(key), string(value)))
end

# ---- the fused hot path: src/GaussDCA.jl:28-42 in one call --------------------------------
function hot_path(Z::Matrix{Int8}, q::Integer, pseudocount::Real, θ, score::Symbol)
    N, M = size(Z)
    S = Matrix{Float64}(undef, N, N)
    p = Ref(GdcaParams(Float64(pseudocount), theta_arg(θ), score == :DI ? 1 : 0, 1))
    st = GdcaStats()
    GC.@preserve Z S begin
        rc = ccall((:gdca_run, libgdca), Cint,
                   (Ptr{Cvoid}, Ptr{Int8}, Int32, Int32, Int32, Ref{GdcaParams}, Ptr{Float64}, Ref{GdcaStats}),
                   ctx(), Z, N, M, q, p, S, st)
    end
    check(rc, st.info)
    return S, st
end

# ... and src/GaussDCA.jl:28-44: the same followed by compute_ranking on the device; only the sorted ranking comes back
function hot_path_ranked(Z::Matrix{Int8}, q::Integer, pseudocount::Real, θ, score::Symbol, min_separation::Integer)
    N, M = size(Z)
    len = max(ccall((:gdca_ranking_length, libgdca), Int64, (Int32, Int32), N, min_separation), 0)
    ri = Vector{Int32}(undef, len)
    rj = Vector{Int32}(undef, len)
    rs = Vector{Float64}(undef, len)
    p = Ref(GdcaParams(Float64(pseudocount), theta_arg(θ), score == :DI ? 1 : 0, 1))
    st = GdcaStats()
    GC.@preserve Z ri rj rs begin
        rc = ccall((:gdca_run_ranked, libgdca), Cint,
                   (Ptr{Cvoid}, Ptr{Int8}, Int32, Int32, Int32, Ref{GdcaParams}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ref{GdcaStats}),
                   ctx(), Z, N, M, q, p, min_separation, ri, rj, rs, st)
    end
    check(rc, st.info)
    return [(Int(ri[t]), Int(rj[t]), rs[t]) for t in 1:len], st
end

function gDCA(filename::AbstractString; pseudocount::Real = 0.8, θ = :auto, max_gap_fraction::Real = 0.9,
              score::Symbol = :frob, min_separation::Integer = 5, remove_dups::Bool = false)
    check_arguments(filename, pseudocount, θ, max_gap_fraction, score, min_separation)
    Z = DCAUtils.read_fasta_alignment(filename, max_gap_fraction)
    if remove_dups
        Z, _ = DCAUtils.remove_duplicate_sequences(Z)
    end
    q = Int(maximum(Z))
    q ≥ 32 && error("parameter q=$q is too big (max 31 is allowed)")
    R, st = hot_path_ranked(Z, q, pseudocount, θ, score, min_separation)
    # (gdca.h, gdca_stats.refined: -1 = the refinement of an ill-conditioned inverse cannot have converged and the Cholesky
    # fallback is switched off -- the Python mirror warns in the same case)
    st.refined < 0 && @warn "covariance too ill-conditioned for the refinement step and option CHOLESKY=0: scores are unreliable" pseudocount kappa_1 = st.matrix_norm1 * st.inverse_norm1
    return R
end

# ---- DCAUtils-named operators (call sites src/GaussDCA.jl:28,30,37,39) ----------------------
function compute_weights(Z::Matrix{Int8}, q::Integer, θ)
    N, M = size(Z)
    W = Vector{Float64}(undef, M); Meff = Ref{Cdouble}(0); th = Ref{Cdouble}(0); thr = Ref{Int32}(0)
    GC.@preserve Z W check(ccall((:gdca_compute_weights, libgdca), Cint,
        (Ptr{Cvoid}, Ptr{Int8}, Int32, Int32, Cdouble, Ptr{Float64}, Ref{Cdouble}, Ref{Cdouble}, Ref{Int32}),
        ctx(), Z, N, M, theta_arg(θ), W, Meff, th, thr))
    return W, Meff[]
end

function compute_weighted_frequencies(Z::Matrix{Int8}, q::Integer, θ)
    W, Meff = compute_weights(Z, q, θ)
    N, M = size(Z); n = N * (q - 1)
    Pi = Vector{Float64}(undef, n); Pij = Matrix{Float64}(undef, n, n)
    GC.@preserve Z W Pi Pij check(ccall((:gdca_frequencies, libgdca), Cint,
        (Ptr{Cvoid}, Ptr{Int8}, Int32, Int32, Int32, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Float64}),
        ctx(), Z, N, M, q, W, Meff, Pi, Pij))
    return Pi, Pij, Meff, W
end

function add_pseudocount(Pi_true::Vector{Float64}, Pij_true::Matrix{Float64}, pc::Float64, q::Integer = 21)
    n = length(Pi_true); N = n ÷ (q - 1)
    Pi = similar(Pi_true); Pij = similar(Pij_true)
    GC.@preserve Pi_true Pij_true Pi Pij check(ccall((:gdca_add_pseudocount, libgdca), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Cdouble, Ptr{Float64}, Ptr{Float64}),
        ctx(), Pi_true, Pij_true, N, q, pc, Pi, Pij))
    return Pi, Pij
end

function inv_cholesky(C::Matrix{Float64})   # mJ = inv(cholesky(C)), src/GaussDCA.jl:34
    A = copy(C); n = size(A, 1); info = Ref{Int32}(0)
    rc = GC.@preserve A ccall((:gdca_spd_inverse, libgdca), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ref{Int32}),
                              ctx(), A, n, info)
    check(rc, info[])
    return A
end

function compute_FN(mJ::Matrix{Float64}, q::Integer = 21)
    N = size(mJ, 1) ÷ (q - 1); S = Matrix{Float64}(undef, N, N)
    GC.@preserve mJ S check(ccall((:gdca_fn, libgdca), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, Ptr{Float64}),
                                  ctx(), mJ, N, q, S))
    return S
end

function compute_DI_gauss(mJ::Matrix{Float64}, C::Matrix{Float64}, q::Integer = 21)
    N = size(mJ, 1) ÷ (q - 1); S = Matrix{Float64}(undef, N, N)
    GC.@preserve mJ C S check(ccall((:gdca_di, libgdca), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ptr{Float64}), ctx(), mJ, C, N, q, S))
    return S
end

# ---- device-resident form of the reference's statement-by-statement pipeline ---------------------------------
# src/GaussDCA.jl:28-42 kept as six statements, every array in HBM (gdca_dbuf handles; `_dev` entry points of
# include/gdca.h): only Z goes in and S comes out over PCIe.
mutable struct DBuf
    h::Ptr{Cvoid}
    function DBuf(bytes::Integer)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:gdca_dbuf_alloc, libgdca), Cint, (Ptr{Cvoid}, UInt64, Ref{Ptr{Cvoid}}), ctx(), bytes, r))
        b = new(r[])
        finalizer(x -> ccall((:gdca_dbuf_free, libgdca), Cint, (Ptr{Cvoid},), x.h), b)
        return b
    end
end
dptr(b::DBuf) = ccall((:gdca_dbuf_ptr, libgdca), Ptr{Cvoid}, (Ptr{Cvoid},), b.h)
upload!(b::DBuf, a::Array) = GC.@preserve a check(ccall((:gdca_dbuf_upload, libgdca), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ptr{Cvoid}, UInt64), ctx(), b.h, 0, a, sizeof(a)))
download!(a::Array, b::DBuf) = GC.@preserve a check(ccall((:gdca_dbuf_download, libgdca), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ptr{Cvoid}, UInt64), ctx(), b.h, 0, a, sizeof(a)))

function gDCA_stepwise(Z::Matrix{Int8}; pseudocount::Real = 0.8, θ = :auto, score::Symbol = :frob,
                       min_separation::Integer = 5)
    N, M = size(Z); q = Int(maximum(Z)); n = N * (q - 1)
    q ≥ 32 && error("parameter q=$q is too big (max 31 is allowed)")
    dZ = DBuf(N * M); upload!(dZ, Z)
    dW = DBuf(8M); dPi = DBuf(8n); dPij = DBuf(8n * n); dS = DBuf(8N * N)
    Meff = Ref{Cdouble}(0); th = Ref{Cdouble}(0); thr = Ref{Int32}(0); info = Ref{Int32}(0)
    c = ctx()
    # Pi_true, Pij_true, Meff, _ = compute_weighted_frequencies(Z, q, θ)                          (:28)
    check(ccall((:gdca_compute_weights_dev, libgdca), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Cdouble, Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Int32}),
        c, dptr(dZ), N, M, theta_arg(θ), dptr(dW), Meff, th, thr))
    check(ccall((:gdca_frequencies_dev, libgdca), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}),
        c, dptr(dZ), N, M, q, dptr(dW), Meff[], dptr(dPi), dptr(dPij)))
    # Pi, Pij = add_pseudocount(Pi_true, Pij_true, Float64(pseudocount), q)   (in place)          (:30)
    check(ccall((:gdca_add_pseudocount_dev, libgdca), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}),
        c, dptr(dPi), dptr(dPij), N, q, Float64(pseudocount), dptr(dPi), dptr(dPij)))
    # C = compute_C(Pi, Pij)                                                                       (:32)
    dC = DBuf(8n * n)
    check(ccall((:gdca_covariance_dev, libgdca), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
        c, dptr(dPi), dptr(dPij), n, dptr(dC)))
    # mJ = inv(cholesky(C))        (dPij is reused for mJ; C itself is kept for compute_DI_gauss)  (:34)
    check(ccall((:gdca_covariance_dev, libgdca), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
        c, dptr(dPi), dptr(dPij), n, dptr(dPij)))
    rc = ccall((:gdca_spd_inverse_dev, libgdca), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Int32}), c, dptr(dPij), n, info)
    check(rc, info[])
    # S = score == :DI ? compute_DI_gauss(mJ, C, q) : compute_FN(mJ, q)                            (:36-40)
    if score == :DI
        check(ccall((:gdca_di_dev, libgdca), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}),
            c, dptr(dPij), dptr(dC), N, q, dptr(dS)))
    else
        check(ccall((:gdca_fn_dev, libgdca), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}),
            c, dptr(dPij), N, q, dptr(dS)))
    end
    # S = correct_APC(S)                                                                           (:42)
    check(ccall((:gdca_apc_dev, libgdca), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32), c, dptr(dS), N))
    S = Matrix{Float64}(undef, N, N); download!(S, dS)
    return compute_ranking(S, min_separation)                                                    # :44
end

end # module
