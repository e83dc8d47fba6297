"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  The product
package (tactile_gym_amd) never does; it fails loudly when its HIP library is missing.
"""
