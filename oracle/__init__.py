"""oracle/ -- CPU restatements of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (text_alignment_amd/) never does.
"""
