# GaussDCAHip.jl -- the reference-side binding of libgdca.so.
#
# Drop-in for the hot path of GaussDCA.jl (src/GaussDCA.jl:28-42): same exported names, same
# keyword arguments and defaults as `gDCA` (src/GaussDCA.jl:8-16); the DCAUtils-named operators
# used by the reference (compute_weights, compute_weighted_frequencies, add_pseudocount,
# compute_FN, compute_DI_gauss) are thin `ccall` wrappers over include/gdca.h.
#
# NOT EXECUTED in the build image (no Julia toolchain there); kept minimal and literal.
module GaussDCAHip

export gDCA, printrank, compute_weights, compute_weighted_frequencies, add_pseudocount,
       compute_FN, compute_DI_gauss

using LinearAlgebra, Printf
import DCAUtils  # host-side I/O only: read_fasta_alignment, remove_duplicate_sequences

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
    ms_total::Cdouble; ms_theta::Cdouble; ms_weights::Cdouble; ms_covariance::Cdouble
    ms_inverse::Cdouble; ms_inverse_update::Cdouble; ms_score::Cdouble
    inverse_flops::Cdouble; update_flops::Cdouble
    GdcaStats() = new()
end

const CTX = Ref{Ptr{Cvoid}}(C_NULL)

function ctx()
    if CTX[] == C_NULL
        dev = parse(Int32, get(ENV, "GDCA_DEVICE", "0"))
        st = ccall((:gdca_ctx_create, libgdca), Cint, (Int32, Ref{Ptr{Cvoid}}), dev, CTX)
        st == 0 || error("gdca_ctx_create failed (status $st): no usable HIP device; there is no CPU fallback")
    end
    return CTX[]
end

function check(st::Integer, info::Integer = 0)
    st == 0 && return
    msg = unsafe_string(ccall((:gdca_last_error, libgdca), Cstring, (Ptr{Cvoid},), ctx()))
    st == 1 && throw(ArgumentError(msg))
    st == 2 && throw(PosDefException(info))
    st == 4 && throw(OutOfMemoryError())
    error("libgdca: $msg")
end

theta_arg(θ) = θ === :auto ? -1.0 : Float64(θ)

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

function gDCA(filename::AbstractString; pseudocount::Real = 0.8, θ = :auto, max_gap_fraction::Real = 0.9,
              score::Symbol = :frob, min_separation::Integer = 5, remove_dups::Bool = false)
    check_arguments(filename, pseudocount, θ, max_gap_fraction, score, min_separation)
    Z = DCAUtils.read_fasta_alignment(filename, max_gap_fraction)
    if remove_dups
        Z, _ = DCAUtils.remove_duplicate_sequences(Z)
    end
    q = Int(maximum(Z))
    q ≥ 32 && error("parameter q=$q is too big (max 31 is allowed)")
    S, _ = hot_path(Z, q, pseudocount, θ, score)
    return compute_ranking(S, min_separation)
end

function check_arguments(filename, pseudocount, θ, max_gap_fraction, score, min_separation)
    aerror(s) = throw(ArgumentError(s))
    0 <= pseudocount <= 1 || aerror("invalid pseudocount value: $pseudocount (must be between 0 and 1)")
    θ == :auto || (θ isa Real && 0 <= θ <= 1) ||
        aerror("invalid θ value: $θ (must be either :auto, or a number between 0 and 1)")
    0 <= max_gap_fraction <= 1 ||
        aerror("invalid max_gap_fraction value: $max_gap_fraction (must be between 0 and 1)")
    score in [:DI, :frob] || aerror("invalid score value: $score (must be either :DI or :frob)")
    min_separation >= 1 || aerror("invalid min_separation value: $min_separation (must be >= 1)")
    isfile(filename) || aerror("cannot open file $filename")
    return true
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

# ---- unchanged host pieces of the reference (src/GaussDCA.jl:67-74, :88-99) ------------------
function compute_ranking(S::Matrix{Float64}, min_separation::Integer = 5)
    N = size(S, 1)
    R = Array{Tuple{Int,Int,Float64}}(undef, ((N - min_separation) * (N - min_separation + 1)) ÷ 2)
    counter = 0
    for i = 1:N-min_separation, j = i+min_separation:N
        counter += 1
        R[counter] = (i, j, S[j, i])
    end
    sort!(R, by = x -> x[3], rev = true)
    return R
end

function printrank(io::IO, R::Vector{Tuple{Int,Int,Float64}})
    for I in R
        @printf(io, "%i %i %e\n", I[1], I[2], I[3])
    end
end
printrank(R::Vector{Tuple{Int,Int,Float64}}) = printrank(stdout, R)
printrank(outfile::AbstractString, R::Vector{Tuple{Int,Int,Float64}}) = open(f -> printrank(f, R), outfile, "w")

end # module
